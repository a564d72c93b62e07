"""GPU parity tests: the HIP path (through the C ABI / host mirror) against the CPU
oracle on the same seeded inputs, against the committed fixtures, and -- at full
BASELINE size -- through size-independent properties.

Tolerances (fp64): per-kernel max|y-y_ref|/max|y_ref| <= 1e-13;
V-cycle output <= 1e-11 relative 2-norm; residual histories <= 1e-8 relative per
entry; iteration counts identical."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import max_rel, rel_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL_KERNEL, TOL_VCYCLE, TOL_HIST = 1e-13, 1e-11, 1e-8


def seeded(n, seed):
    return np.random.Generator(np.random.MT19937(seed)).uniform(-1.0, 1.0, size=n)


def jac(S, nlev, niter=10, omega=2.0 / 3.0):
    sm = S.RichardsonSmoother(S.JacobiLinearSolver(), niter, omega)
    return [sm] * (nlev - 1)


def make_gmg(S, H, **kw):
    nlev = len(H["mats"])
    kw.setdefault("pre_smoothers", jac(S, nlev))
    kw.setdefault("post_smoothers", kw["pre_smoothers"])
    kw.setdefault("maxiter", 1)
    return S.GMGLinearSolver(H["mats"], H["prolongations"], kw.pop("restrict", H["restrictions"]), **kw)


def setup(S, solver, A):
    return S.numerical_setup(S.symbolic_setup(solver, A), A)


# ---------------------------------------------------------------- per-kernel parity
@pytest.mark.parametrize("nc,nlev,order", [((64, 64), 3, 1), ((16, 16, 16), 3, 1), ((12, 20, 8), 2, 1), ((8, 8, 8), 2, 2)])
def test_operator_apply_parity(S, orc, hierarchy, nc, nlev, order):
    """K1/K4/K5: mul!(y,A,x), mul!(rH,R,rh), mul!(dxh,P,dxH) on every level."""
    H = hierarchy(nc, nlev, order)
    ns = setup(S, make_gmg(S, H), H["mats"][0])
    from gridapsolvers_jl_amd import abi
    for l in range(nlev):
        A = H["mats"][l]
        x = seeded(A.shape[0], 100 + l)
        y = np.zeros(A.shape[0])
        ns.op_apply(l, abi.OP_A, x, y)
        assert max_rel(y, orc.spmv(A, x)) <= TOL_KERNEL
        if l < nlev - 1:
            P, R = H["prolongations"][l], H["restrictions"][l]
            xc = seeded(P.shape[1], 200 + l)
            yp = np.zeros(P.shape[0]); ns.op_apply(l, abi.OP_P, xc, yp)
            assert max_rel(yp, orc.spmv(P, xc)) <= TOL_KERNEL
            yr = np.zeros(R.shape[0]); ns.op_apply(l, abi.OP_R, x, yr)
            assert max_rel(yr, orc.spmv(R, x)) <= TOL_KERNEL


@pytest.mark.parametrize("niter,omega", [(1, 1.0), (3, 0.5), (10, 2.0 / 3.0)])
def test_richardson_jacobi_sweeps_parity(S, orc, hierarchy, niter, omega):
    """K2+K1+K3 fused sweep vs RichardsonSmoothers.jl:84-98 executed literally by the oracle (odd and even niter)."""
    nc, nlev = (16, 16, 16), 3
    H = hierarchy(nc, nlev)
    ns = setup(S, make_gmg(S, H, pre_smoothers=jac(S, nlev, niter, omega)), H["mats"][0])
    go = orc.GMG(H["mats"], H["prolongations"], pre_smoothers=[orc.Smoother(orc.JACOBI, niter, omega)] * (nlev - 1), maxiter=1)
    for l in range(nlev - 1):
        n = H["mats"][l].shape[0]
        x0, r0 = seeded(n, 1 + l), seeded(n, 50 + l)
        x, r = x0.copy(), r0.copy()
        ns.smooth(l, x, r)
        xo, ro = go.smooth(l, x0, r0)
        assert max_rel(x, xo) <= TOL_KERNEL and max_rel(r, ro) <= TOL_KERNEL


def test_jacobi_precond_coarse_solve_and_dot(S, orc, hierarchy):
    nc, nlev = (16, 16, 16), 3
    H = hierarchy(nc, nlev)
    ns = setup(S, make_gmg(S, H), H["mats"][0])
    go = orc.GMG(H["mats"], H["prolongations"], maxiter=1)
    n = H["mats"][0].shape[0]
    r = seeded(n, 3)
    dx = np.zeros(n); ns.precond(0, r, dx)
    assert max_rel(dx, go.precond(0, r)) <= 1e-15          # inv_diag .* r: bitwise-level
    nL = H["mats"][-1].shape[0]
    rc = seeded(nL, 4); xc = np.zeros(nL)
    ns.coarse_solve(rc, xc)
    assert max_rel(xc, go.coarse_solve(rc)) <= 1e-12        # dense inverse GEMV vs banded LU (kappa ~ 1e1)
    a, b = seeded(n, 5), seeded(n, 6)
    assert abs(ns.dot(a, b) - orc.dot(a, b)) <= 1e-13 * np.sqrt(n)
    assert abs(ns.dot(a[:-1], b[:-1]) - orc.dot(a[:-1].copy(), b[:-1].copy())) <= 1e-13 * np.sqrt(n)   # odd length


# ---------------------------------------------------------------- V / W / F cycles, solver mode
def test_vcycle_golden_config1(S, hierarchy):
    gold = np.load(os.path.join(GOLD, "config1_q1_64x64.npz"))
    H = hierarchy((64, 64), 3)
    gmg = make_gmg(S, H)
    ns = setup(S, gmg, H["mats"][0])
    n = H["mats"][0].shape[0]
    r = seeded(n, 7)
    z = np.full(n, 123.0)                                    # must be overwritten (fill!(x,0), GMGLinearSolvers.jl:619)
    S.solve_(z, ns, r)
    assert rel_err(z, gold["vcycle_z"]) <= TOL_VCYCLE
    assert gmg.log.num_iters == 1
    np.testing.assert_allclose(gmg.log.residuals[:2], gold["vcycle_hist"], rtol=TOL_HIST)
    x, rr = np.zeros(n), r.copy()
    ns.smooth(0, x, rr)
    assert max_rel(x, gold["smooth_x"]) <= TOL_KERNEL and max_rel(rr, gold["smooth_r"]) <= TOL_KERNEL


@pytest.mark.parametrize("cyc", ["v", "w", "f"])
def test_cycle_types_golden(S, hierarchy, cyc):
    """gmg_v_cycle!/gmg_w_cycle!/gmg_f_cycle! (GMGLinearSolvers.jl:468-610)."""
    gold = np.load(os.path.join(GOLD, "q1_16cubed.npz"))
    H = hierarchy((16, 16, 16), 3)
    gmg = make_gmg(S, H, cycle_type=f"{cyc}_cycle")
    ns = setup(S, gmg, H["mats"][0])
    r = seeded(H["mats"][0].shape[0], 11)
    z = np.zeros_like(r)
    S.solve_(z, ns, r)
    assert rel_err(z, gold[f"z_{cyc}"]) <= TOL_VCYCLE
    np.testing.assert_allclose(gmg.log.residuals[:2], gold[f"h_{cyc}"], rtol=TOL_HIST)


def test_solver_mode_golden(S, po, hierarchy):
    """mode=:solver (GMGLinearSolvers.jl:621-625) iterated to rtol 1e-8."""
    gold = np.load(os.path.join(GOLD, "q1_16cubed.npz"))
    nc = (16, 16, 16)
    H = hierarchy(nc, 3)
    gmg = make_gmg(S, H, mode="solver", maxiter=6, atol=1e-14, rtol=1e-8)
    ns = setup(S, gmg, H["mats"][0])
    b = po.dirichlet_lift_rhs(nc, 1)
    x = np.zeros_like(b)
    S.solve_(x, ns, b)
    assert gmg.log.num_iters == int(gold["solver_niters"]) and gmg.log.flag == int(gold["solver_flag"])
    np.testing.assert_allclose(gmg.log.residuals[: gmg.log.num_iters + 1], gold["solver_hist"], rtol=TOL_HIST)
    assert rel_err(x, gold["solver_x"]) <= 1e-10
    assert po.l2_error_sq(nc, 1, x) < 1e-8


# ---------------------------------------------------------------- outer Krylov solvers
@pytest.mark.parametrize("nc,nlev", [((64, 64), 3), ((32, 32, 32), 3), ((16, 24, 8), 2)])
def test_cg_gmg_matches_oracle(S, po, orc, hierarchy, nc, nlev):
    """BASELINE configs 1/2 shape: CG(maxiter=20,atol=1e-14,rtol=1e-6) + GMG V-cycle preconditioner."""
    H = hierarchy(nc, nlev)
    b = po.dirichlet_lift_rhs(nc, 1)
    solver = S.CGSolver(make_gmg(S, H), maxiter=20, atol=1e-14, rtol=1e-6)
    ns = setup(S, solver, H["mats"][0])
    x = np.zeros_like(b)
    S.solve_(x, ns, b)
    g = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=1)
    xo, nit, flag, hist = orc.cg_solve(H["mats"][0], b, Pl=g, maxiter=20, atol=1e-14, rtol=1e-6)
    assert solver.log.num_iters == nit and solver.log.flag == flag
    np.testing.assert_allclose(solver.log.residuals[: nit + 1], hist, rtol=TOL_HIST)
    assert rel_err(x, xo) <= 1e-10
    assert po.l2_error_sq(nc, 1, x) < 1e-8                   # the reference tests' own criterion


def test_cg_golden_config1_and_device_tensors(S, po, hierarchy):
    import torch
    gold = np.load(os.path.join(GOLD, "config1_q1_64x64.npz"))
    nc = (64, 64)
    H = hierarchy(nc, 3)
    b = po.dirichlet_lift_rhs(nc, 1)
    solver = S.CGSolver(make_gmg(S, H), maxiter=20, atol=1e-14, rtol=1e-6)
    ns = setup(S, solver, H["mats"][0])
    xd = torch.zeros(b.size, dtype=torch.float64, device="cuda")
    bd = torch.from_numpy(b).cuda()
    torch.cuda.synchronize()
    S.solve_(xd, ns, bd)
    assert solver.log.num_iters == int(gold["niters"]) == 3
    np.testing.assert_allclose(solver.log.residuals[:4], gold["hist"], rtol=TOL_HIST)
    assert rel_err(xd.cpu().numpy(), gold["x"]) <= 1e-10
    assert torch.equal(bd.cpu(), torch.from_numpy(b))        # b untouched
    # run-to-run determinism (fixed reduction trees, no atomics)
    xd2 = torch.zeros_like(xd)
    S.solve_(xd2, ns, bd)
    assert torch.equal(xd, xd2)


@pytest.mark.parametrize("flexible", [False, True])
@pytest.mark.parametrize("pc", ["none", "jacobi"])
def test_cg_variants_reference_krylov_tests(S, po, orc, hierarchy, flexible, pc):
    """KrylovTests.jl:77-90: CGSolver(), CGSolver(Jacobi), flexible CG; criterion E < 1e-6."""
    nc = (8, 8, 8)
    H = hierarchy(nc, 2)
    b = po.dirichlet_lift_rhs(nc, 1)
    gmg = make_gmg(S, H)
    P = (None, gmg) if pc == "none" else (S.JacobiLinearSolver(), gmg)
    solver = S.CGSolver(P, rtol=1e-8, flexible=flexible)
    ns = setup(S, solver, H["mats"][0])
    x = np.zeros_like(b)
    S.solve_(x, ns, b)
    xo, nit, flag, hist = orc.cg_solve(H["mats"][0], b, Pl=None if pc == "none" else "jacobi", rtol=1e-8, flexible=flexible)
    assert solver.log.num_iters == nit and solver.log.flag == flag
    np.testing.assert_allclose(solver.log.residuals[: nit + 1], hist, rtol=1e-6)
    assert po.l2_error_sq(nc, 1, x) < 1e-6


@pytest.mark.parametrize("m,restart", [(5, False), (2, True), (2, False)])
def test_fgmres_gmg_matches_oracle(S, po, orc, hierarchy, m, restart):
    """FGMRESSolvers.jl:130-199 incl. restart and basis growth (m_add)."""
    nc, nlev = (32, 32), 3
    H = hierarchy(nc, nlev)
    b = po.dirichlet_lift_rhs(nc, 1)
    solver = S.FGMRESSolver(m, make_gmg(S, H, cycle_type="w_cycle"), restart=restart, maxiter=20, atol=1e-14, rtol=1e-9)
    ns = setup(S, solver, H["mats"][0])
    x = np.zeros_like(b)
    S.solve_(x, ns, b)
    g = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], cycle=orc.W_CYCLE, maxiter=1)
    xo, nit, flag, hist = orc.fgmres_solve(H["mats"][0], b, Pr=g, m=m, restart=restart, maxiter=20, atol=1e-14, rtol=1e-9)
    assert solver.log.num_iters == nit and solver.log.flag == flag
    np.testing.assert_allclose(solver.log.residuals[: nit + 1], hist, rtol=1e-6)
    assert rel_err(x, xo) <= 1e-9


# ---------------------------------------------------------------- patch smoother (config 3 shape)
@pytest.mark.parametrize("kind", ["patch", "block"])
def test_q2_patch_smoother_parity(S, po, orc, hierarchy, kind):
    """Richardson(PatchSolver|BlockJacobiSolver,10,0.2) on Q2 (GMGTests.jl:18-47) + FGMRES(5)."""
    gold = np.load(os.path.join(GOLD, "q2_16x16_patch.npz"))
    nc, nlev, order = (16, 16), 2, 2
    H = hierarchy(nc, nlev, order)
    pp, pd = po.vertex_star_patches(nc, order)
    M = S.PatchSolver(pp, pd) if kind == "patch" else S.BlockJacobiSolver(pp, pd)
    sm = [S.RichardsonSmoother(M, 10, 0.2)]
    solver = S.FGMRESSolver(5, make_gmg(S, H, pre_smoothers=sm), maxiter=20, atol=1e-14, rtol=1e-6)
    ns = setup(S, solver, H["mats"][0])
    b = po.dirichlet_lift_rhs(nc, order)
    r = seeded(b.size, 13)
    dx = np.zeros_like(r)
    ns.P_ns.precond(0, r, dx)
    assert max_rel(dx, gold["patch_dx"]) <= 1e-12
    x = np.zeros_like(b)
    S.solve_(x, ns, b)
    assert solver.log.num_iters == int(gold["niters"])
    np.testing.assert_allclose(solver.log.residuals[: solver.log.num_iters + 1], gold["hist"], rtol=1e-6)
    assert rel_err(x, gold["x"]) <= 1e-9
    assert po.l2_error_sq(nc, order, x) < 1e-8


def test_q2_3d_patch_smoother_vs_oracle(S, po, orc, hierarchy):
    nc, nlev, order = (8, 8, 8), 2, 2
    H = hierarchy(nc, nlev, order)
    pp, pd = po.vertex_star_patches(nc, order)
    assert (pp[1:] - pp[:-1]).max() == 27                    # n_p = 27 in 3-D
    gmg = make_gmg(S, H, pre_smoothers=[S.RichardsonSmoother(S.PatchSolver(pp, pd), 10, 0.2)])
    ns = setup(S, gmg, H["mats"][0])
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=[orc.Smoother(orc.PATCH, 10, 0.2, pp, pd)], maxiter=1)
    r = seeded(H["mats"][0].shape[0], 17)
    z = np.zeros_like(r)
    S.solve_(z, ns, r)
    zo, _, _, ho = go.solve(r)
    assert rel_err(z, zo) <= TOL_VCYCLE
    np.testing.assert_allclose(gmg.log.residuals[:2], ho, rtol=TOL_HIST)


# ---------------------------------------------------------------- input formats & edge cases
def test_input_formats_csc_one_based_int64(S, po, orc, hierarchy, pkg):
    """The ABI accepts {CSR|CSC} x {0|1}-based x {Int32|Int64} (Julia SparseMatrixCSC{Float64,Int64} is CSC/1/8)."""
    from gridapsolvers_jl_amd import abi
    nc, nlev = (12, 10, 6), 2
    H = hierarchy(nc, nlev)

    class Mat:
        pass

    def julia_csc(M):
        csc = M.to_scipy().tocsc(); csc.sort_indices()
        o = Mat(); o.shape = csc.shape; o.ptr = csc.indptr.astype(np.int64) + 1; o.idx = csc.indices.astype(np.int64) + 1
        o.val = csc.data.copy(); o.layout = abi.CSC; o.index_base = 1
        return o

    Hj = dict(mats=[julia_csc(A) for A in H["mats"]], prolongations=[julia_csc(P) for P in H["prolongations"]],
              restrictions=[None] * (nlev - 1))                  # R = P^T built by the library
    r = seeded(H["mats"][0].shape[0], 23)
    z1, z2 = np.zeros_like(r), np.zeros_like(r)
    S.solve_(z1, setup(S, make_gmg(S, H), H["mats"][0]), r)
    S.solve_(z2, setup(S, make_gmg(S, Hj), Hj["mats"][0]), r)
    assert rel_err(z2, z1) <= 1e-14
    zo = orc.GMG(H["mats"], H["prolongations"], maxiter=1).solve(r)[0]
    assert rel_err(z1, zo) <= TOL_VCYCLE


def test_ragged_empty_and_long_rows(S, orc, po, pkg):
    """CSR edge cases of the stream kernel: empty rows, rows longer than the LDS tile (2048 nnz), 1-nnz rows."""
    from gridapsolvers_jl_amd import abi
    rng = np.random.default_rng(99)
    n = 5000
    lens = rng.integers(0, 40, size=n)
    lens[10] = 0; lens[11] = 0; lens[n - 1] = 0
    lens[100] = 3000; lens[101] = 2048; lens[102] = 2049; lens[4000] = 4999
    ptr = np.zeros(n + 1, dtype=np.int64); np.cumsum(lens, out=ptr[1:])
    idx = np.concatenate([np.sort(rng.choice(n, size=l, replace=False)) for l in lens]).astype(np.int32)
    val = rng.uniform(-1, 1, size=idx.size)
    # make it usable as a "level matrix": add a dominant diagonal through a second matrix A = B + D
    import scipy.sparse as sp
    B = sp.csr_matrix((val, idx, ptr), shape=(n, n))
    A = po.CSR((n, n), ptr, idx, val)
    Asq = (B + sp.diags(np.full(n, 100.0))).tocsr(); Asq.sort_indices()
    Afull = po.CSR((n, n), Asq.indptr, Asq.indices, Asq.data)
    nH = 50
    Pm = sp.csr_matrix((np.ones(n), (np.arange(n), np.arange(n) % nH)), shape=(n, nH)); Pm.sort_indices()
    P = po.CSR((n, nH), Pm.indptr, Pm.indices, Pm.data)
    AH = (Pm.T @ Asq @ Pm).tocsr(); AH.sort_indices()
    H = dict(mats=[Afull, po.CSR((nH, nH), AH.indptr, AH.indices, AH.data)], prolongations=[P], restrictions=[P.transpose()])
    ns = setup(S, make_gmg(S, H), Afull)
    x = seeded(n, 1)
    y = np.zeros(n)
    ns.op_apply(0, abi.OP_A, x, y)
    assert max_rel(y, orc.spmv(Afull, x)) <= TOL_KERNEL
    yr = np.zeros(nH); ns.op_apply(0, abi.OP_R, x, yr)        # R rows have n/nH = 100 nnz
    assert max_rel(yr, orc.spmv(H["restrictions"][0], x)) <= TOL_KERNEL
    xs, rs = np.zeros(n), x.copy()
    ns.smooth(0, xs, rs)
    xo, ro = orc.GMG(H["mats"], H["prolongations"], maxiter=1).smooth(0, np.zeros(n), x)
    assert max_rel(xs, xo) <= TOL_KERNEL and max_rel(rs, ro) <= TOL_KERNEL
    del A


def test_error_behaviour_on_device(S, po, hierarchy, pkg):
    from gridapsolvers_jl_amd import abi
    H = hierarchy((8, 8), 2)
    gmg = make_gmg(S, H)
    ns = setup(S, gmg, H["mats"][0])
    with pytest.raises(ValueError):
        S.solve_(np.zeros(3), ns, np.zeros(3))                   # wrong length
    lib = abi.load()
    h = C.c_void_p()
    assert lib.gmg_create(C.byref(h), 2, 0) == abi.OK
    n = 49
    x = np.zeros(n)
    assert lib.gmg_apply(h, x.ctypes.data, x.ctypes.data, abi.MEM_HOST, None, None, 0) == abi.ERR_STATE   # before setup
    assert b"gmg_setup" in lib.gmg_last_error(h)
    assert lib.gmg_setup(h) == abi.ERR_STATE                     # matrices missing
    # singular coarse matrix -> GMG_ERR_SINGULAR
    Z = po.CSR((4, 4), np.arange(5), np.arange(4), np.zeros(4))
    bad = dict(mats=[H["mats"][0], H["mats"][1]], prolongations=H["prolongations"], restrictions=H["restrictions"])
    bad["mats"][1] = po.CSR(H["mats"][1].shape, H["mats"][1].ptr, H["mats"][1].idx, np.zeros_like(H["mats"][1].val))
    with pytest.raises(abi.GmgError) as ei:
        setup(S, make_gmg(S, bad), bad["mats"][0])
    assert ei.value.code == abi.ERR_SINGULAR
    lib.gmg_destroy(h)
    del Z


def test_maxiter_and_tolerance_flags(S, po, orc, hierarchy):
    """SolverTolerances.jl:97-128 through the device CG."""
    nc = (16, 16)
    H = hierarchy(nc, 3)
    b = po.dirichlet_lift_rhs(nc, 1)
    s1 = S.CGSolver(make_gmg(S, H), maxiter=2, atol=0.0, rtol=1e-30)
    x = np.zeros_like(b); S.solve_(x, setup(S, s1, H["mats"][0]), b)
    assert s1.log.num_iters == 2 and s1.log.flag == S.SOLVER_DIVERGED_MAXITER
    s2 = S.CGSolver(make_gmg(S, H), maxiter=50, atol=1e-12, rtol=1e-6)
    x = np.ones_like(b); S.solve_(x, setup(S, s2, H["mats"][0]), np.zeros_like(b))   # b = 0, x0 != 0
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=1)
    xo, nit, flag, hist = orc.cg_solve(H["mats"][0], np.zeros_like(b), Pl=go, x0=np.ones_like(b), maxiter=50, atol=1e-12, rtol=1e-6)
    assert (s2.log.num_iters, s2.log.flag) == (nit, flag) and np.linalg.norm(x - xo) <= 1e-9 * np.sqrt(b.size)
    s3 = S.CGSolver(make_gmg(S, H), maxiter=50, atol=1e-12, rtol=1e-6)
    x = np.zeros_like(b); S.solve_(x, setup(S, s3, H["mats"][0]), np.zeros_like(b))  # zero rhs: atol at init
    assert s3.log.num_iters == 0 and s3.log.flag == S.SOLVER_CONVERGED_ATOL and np.all(x == 0)


# ---------------------------------------------------------------- BASELINE config 2 at full size: properties
def test_config2_full_size_properties(S, po, hierarchy):
    """3-D Q1 128^3, 4 levels (2 048 383 dofs, 54.4 M nnz): size-independent checks --
    analytic solution reached within the reference's criterion, iteration count equal to the
    oracle's count at every smaller size (3), residual history within 5% of BASELINE.md's
    survey-time values, linearity of the V-cycle, and an exact-residual check."""
    import torch
    nc, nlev = (128, 128, 128), 4
    H = hierarchy(nc, nlev)
    A = H["mats"][0]
    assert A.shape[0] == 2048383 and A.nnz == 54439939
    b = po.dirichlet_lift_rhs(nc, 1)
    gmg = make_gmg(S, H)
    solver = S.CGSolver(gmg, maxiter=20, atol=1e-14, rtol=1e-6)
    ns = setup(S, solver, A)
    xd = torch.zeros(b.size, dtype=torch.float64, device="cuda")
    bd = torch.from_numpy(b).cuda()
    torch.cuda.synchronize()
    S.solve_(xd, ns, bd)
    assert solver.log.num_iters == 3 and solver.log.flag == S.SOLVER_CONVERGED_RTOL
    h = solver.log.residuals[:4] / solver.log.residuals[0]
    np.testing.assert_allclose(h, [1, 7.4e-3, 3.2e-5, 1.9e-7], rtol=0.05)
    x = xd.cpu().numpy()
    assert po.l2_error_sq(nc, 1, x) < 1e-8
    # true residual agrees with the recurrence residual CG reports
    r_true = np.linalg.norm(b - A.matvec(x))
    assert abs(r_true - solver.log.residuals[3]) <= 1e-6 * solver.log.residuals[0]
    # V-cycle is a linear operator: M(a r1 + r2) == a M r1 + M r2
    g = ns.P_ns
    r1, r2 = torch.from_numpy(seeded(b.size, 1)).cuda(), torch.from_numpy(seeded(b.size, 2)).cuda()
    z1, z2, z3 = torch.zeros_like(r1), torch.zeros_like(r1), torch.zeros_like(r1)
    torch.cuda.synchronize()
    S.solve_(z1, g, r1); S.solve_(z2, g, r2)
    r3 = (0.75 * r1 + r2).contiguous(); torch.cuda.synchronize()
    S.solve_(z3, g, r3)
    assert (torch.linalg.norm(z3 - (0.75 * z1 + z2)) / torch.linalg.norm(z3)).item() < 1e-12
    # symmetry of the preconditioner (needed by CG): <M r1, r2> == <r1, M r2>
    a12, a21 = torch.dot(z1, r2).item(), torch.dot(r1, z2).item()
    assert abs(a12 - a21) <= 1e-10 * max(abs(a12), abs(a21))


# ---------------------------------------------------------------- storage formats of the operator stream
def test_storage_formats_agree_bitwise(S, po, orc, hierarchy, monkeypatch):
    """CSR-stream, SELL-64 and compressed SELL-C (16-bit column offsets + 8-bit value dictionary) are
    different LAYOUTS of the same operator.  SELL and SELL-C sum every row left to right over the
    bit-identical decoded (col,val) pairs, so their results must agree to the last bit; the CSR-stream
    kernel reduces with a lane tree and agrees to rounding.  All must match the oracle."""
    nc, nlev = (24, 20, 16), 3
    H = hierarchy(nc, nlev)
    b = po.dirichlet_lift_rhs(nc, 1)
    r = seeded(b.size, 41)
    out = {}
    for name, env in {"csr": dict(GMG_SELL="0"), "sell": dict(GMG_SELL="1", GMG_VDICT="0", GMG_IDX16="0", GMG_PATTERN="0", GMG_OPATTERN="0"),
                      "sell_offsets": dict(GMG_SELL="1", GMG_VDICT="0", GMG_IDX16="0", GMG_PATTERN="0", GMG_OPATTERN="1"),
                      "sellc": dict(GMG_SELL="1", GMG_VDICT="1", GMG_IDX16="1", GMG_PATTERN="0"),
                      "sell_idx16": dict(GMG_SELL="1", GMG_VDICT="0", GMG_IDX16="2", GMG_PATTERN="0"),
                      "sell_dict": dict(GMG_SELL="1", GMG_VDICT="1", GMG_IDX16="0", GMG_PATTERN="0"),
                      "pattern_shared": dict(GMG_PATTERN="1", GMG_PAT_SHARED="1"),
                      "pattern_shared_rb9": dict(GMG_PATTERN="1", GMG_PAT_SHARED="1", GMG_PAT_RB="9"),
                      "pattern_shared_strict": dict(GMG_PATTERN="1", GMG_PAT_SHARED="1", GMG_PAT_STRICT="1"),
                      "pattern_shared_unmasked": dict(GMG_PATTERN="1", GMG_PAT_SHARED="1", GMG_PAT_STRICT="0"),
                      "pattern_shared_unmasked_sweeps": dict(GMG_PATTERN="1", GMG_PAT_SHARED="1", GMG_PAT_STRICT="0", GMG_PERSIST="0"),
                      "pattern_shared_sweeps": dict(GMG_PATTERN="1", GMG_PAT_SHARED="1", GMG_PERSIST="0"),
                      "pattern_shared_strict_sweeps": dict(GMG_PATTERN="1", GMG_PAT_SHARED="1", GMG_PAT_STRICT="1", GMG_PERSIST="0"),
                      "pattern_generic": dict(GMG_PATTERN="1", GMG_PAT_SHARED="0"),
                      "pattern_generic_un3": dict(GMG_PATTERN="1", GMG_PAT_SHARED="0", GMG_PAT_UN="3"),
                      "pattern_two_gather": dict(GMG_PATTERN="1", GMG_ONE_GATHER="0")}.items():
        for k in ("GMG_SELL", "GMG_VDICT", "GMG_IDX16", "GMG_PATTERN", "GMG_PAT_SHARED", "GMG_PAT_RB", "GMG_PAT_UN", "GMG_ONE_GATHER", "GMG_OPATTERN", "GMG_PAT_STRICT",
                  "GMG_PERSIST"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        solver = S.CGSolver(make_gmg(S, H), maxiter=20, atol=1e-14, rtol=1e-6)
        ns = setup(S, solver, H["mats"][0])
        x = np.zeros_like(b)
        S.solve_(x, ns, b)
        z = np.zeros_like(r)
        S.solve_(z, ns.P_ns, r)
        out[name] = (x, z, solver.log.num_iters, solver.log.residuals[: solver.log.num_iters + 1].copy())
        assert ns.P_ns.level_format(0)["layout"] == ("CSR-stream" if name == "csr" else "SELL-P" if name.startswith("pattern") else
                                                    "SELL-O" if name == "sell_offsets" else "SELL-64"), name
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=1)
    xo, nit, flag, hist = orc.cg_solve(H["mats"][0], b, Pl=go, maxiter=20, atol=1e-14, rtol=1e-6)
    zo = go.solve(r)[0]
    for name, (x, z, it, h) in out.items():
        assert it == nit, name
        assert rel_err(x, xo) <= 1e-10 and rel_err(z, zo) <= TOL_VCYCLE, name
        np.testing.assert_allclose(h, hist, rtol=TOL_HIST)
    # (the row-pattern kernels add explicit +0.0 terms for absent entries: the bits of the sums do not change)
    for name in ("sell_offsets", "sellc", "sell_idx16", "sell_dict", "pattern_shared", "pattern_shared_rb9", "pattern_shared_strict",
                 "pattern_shared_unmasked", "pattern_shared_unmasked_sweeps",
                 "pattern_shared_sweeps", "pattern_shared_strict_sweeps", "pattern_generic", "pattern_generic_un3"):
        assert np.array_equal(out[name][0], out["sell"][0]) and np.array_equal(out[name][1], out["sell"][1]), name


def test_value_dictionary_falls_back_on_many_distinct_values(S, po, orc, hierarchy, monkeypatch):
    """A matrix with > 256 distinct values must take the uncompressed path and still be exact."""
    from gridapsolvers_jl_amd import abi
    nc, nlev = (12, 12, 12), 2
    H = hierarchy(nc, nlev)
    A = H["mats"][0]
    rng = np.random.default_rng(8)
    D = rng.uniform(0.5, 2.0, A.shape[0])                       # symmetric diagonal scaling: all values distinct
    As = A.to_scipy().multiply(D[:, None]).multiply(D[None, :]).tocsr(); As.sort_indices()
    A2 = po.CSR(A.shape, As.indptr, As.indices, As.data)
    H2 = dict(mats=[A2, H["mats"][1]], prolongations=H["prolongations"], restrictions=H["restrictions"])
    ns = setup(S, make_gmg(S, H2), A2)
    x = seeded(A.shape[0], 3)
    y = np.zeros_like(x)
    ns.op_apply(0, abi.OP_A, x, y)
    assert max_rel(y, orc.spmv(A2, x)) <= 1e-15                 # sequential row sums: bit-level agreement


def test_numerical_setup_bang_updates_values(S, po, orc, hierarchy):
    """numerical_setup!(ns, A) (CGSolvers.jl:57-63 -> GMGLinearSolvers.jl:249-297): same pattern, new values.
    Scaling the finest operator by 2 must halve the solution of A x = b."""
    nc, nlev = (16, 16), 3
    H = hierarchy(nc, nlev)
    b = po.dirichlet_lift_rhs(nc, 1)
    solver = S.CGSolver(make_gmg(S, H), maxiter=30, atol=1e-14, rtol=1e-10)
    ns = setup(S, solver, H["mats"][0])
    x1 = np.zeros_like(b); S.solve_(x1, ns, b)
    A2 = po.CSR(H["mats"][0].shape, H["mats"][0].ptr, H["mats"][0].idx, 2.0 * H["mats"][0].val)
    S.numerical_setup_(ns, A2)
    x2 = np.zeros_like(b); S.solve_(x2, ns, b)
    assert rel_err(2.0 * x2, x1) < 1e-8
    H2 = dict(mats=[A2] + H["mats"][1:], prolongations=H["prolongations"], restrictions=H["restrictions"])
    go = orc.GMG(H2["mats"], H2["prolongations"], H2["restrictions"], maxiter=1)
    xo, nit, _, _ = orc.cg_solve(A2, b, Pl=go, maxiter=30, atol=1e-14, rtol=1e-10)
    assert solver.log.num_iters == nit and rel_err(x2, xo) < 1e-10


def test_int64_row_pointer_path(S, po, orc, hierarchy, monkeypatch):
    """Matrices with >= 2^31 stored entries (config 3: 8.5e9 nnz) need 64-bit row pointers; the code path is
    forced here on a small hierarchy (CSR-stream kernels, D^-1 extraction and patch-block extraction all take
    the pointer type as a template parameter)."""
    monkeypatch.setenv("GMG_FORCE_PTR64", "1")
    monkeypatch.setenv("GMG_SELL", "0")
    nc, nlev, order = (8, 8, 8), 2, 2
    H = hierarchy(nc, nlev, order)
    pp, pd = po.vertex_star_patches(nc, order)
    gmg = make_gmg(S, H, pre_smoothers=[S.RichardsonSmoother(S.PatchSolver(pp, pd), 10, 0.2)])
    ns = setup(S, gmg, H["mats"][0])
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=[orc.Smoother(orc.PATCH, 10, 0.2, pp, pd)], maxiter=1)
    r = seeded(H["mats"][0].shape[0], 19)
    z = np.zeros_like(r)
    S.solve_(z, ns, r)
    assert rel_err(z, go.solve(r)[0]) <= TOL_VCYCLE
    H1 = hierarchy((16, 16, 16), 3)
    b = po.dirichlet_lift_rhs((16, 16, 16), 1)
    solver = S.CGSolver(make_gmg(S, H1), maxiter=20, atol=1e-14, rtol=1e-6)
    x = np.zeros_like(b); S.solve_(x, setup(S, solver, H1["mats"][0]), b)
    g1 = orc.GMG(H1["mats"], H1["prolongations"], H1["restrictions"], maxiter=1)
    xo, nit, _, hist = orc.cg_solve(H1["mats"][0], b, Pl=g1, maxiter=20, atol=1e-14, rtol=1e-6)
    assert solver.log.num_iters == nit and rel_err(x, xo) <= 1e-10


def test_sellc_mixed_slices_wide_and_banded(S, po, orc, monkeypatch):
    """SELL-C decides 16-bit offsets PER SLICE: a matrix whose first half is banded (offsets fit) and whose
    second half has columns spread over > 65536 (they do not) with few distinct values (dictionary on) must go
    through both decode paths of one kernel and stay exact; also rows of unequal length (masked padding)."""
    from gridapsolvers_jl_amd import abi
    import scipy.sparse as sp
    rng = np.random.default_rng(123)
    n = 150000
    vals = np.array([1.0, -0.5, 2.0, 0.125, -3.0])
    rows, cols, data = [], [], []
    for i in range(n):
        k = int(rng.integers(3, 12))
        if i < n // 2:
            c = np.unique(np.clip(i + rng.integers(-40, 41, size=k), 0, n - 1))       # banded half
        else:
            c = np.unique(rng.integers(0, n, size=k))                                    # wide half
        rows.append(np.full(c.size, i)); cols.append(c); data.append(vals[rng.integers(0, vals.size, size=c.size)])
    B = sp.csr_matrix((np.concatenate(data), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
    B = (B + sp.diags(np.full(n, 64.0))).tocsr(); B.sort_indices()
    A = po.CSR((n, n), B.indptr, B.indices, B.data)
    nH = 97
    Pm = sp.csr_matrix((np.ones(n), (np.arange(n), np.arange(n) % nH)), shape=(n, nH)); Pm.sort_indices()
    P = po.CSR((n, nH), Pm.indptr, Pm.indices, Pm.data)
    AH = (Pm.T @ B @ Pm).tocsr(); AH.sort_indices()
    H = dict(mats=[A, po.CSR((nH, nH), AH.indptr, AH.indices, AH.data)], prolongations=[P], restrictions=[P.transpose()])
    monkeypatch.setenv("GMG_SELL_MAXPAD", "3.0")
    monkeypatch.setenv("GMG_PATTERN", "0")
    ns = setup(S, make_gmg(S, H), A)
    fmt = ns.level_format(0)
    assert fmt["layout"] == "SELL-64" and fmt["value_dictionary"] and fmt["idx16"]
    assert 2.0 < fmt["stream_bytes_per_nnz"] - 1.0 < 4.0          # a mix of 2-byte and 4-byte column slices
    x = seeded(n, 77)
    y = np.zeros(n)
    ns.op_apply(0, abi.OP_A, x, y)
    assert max_rel(y, orc.spmv(A, x)) <= 1e-15
    xs, rs = np.zeros(n), x.copy()
    ns.smooth(0, xs, rs)
    xo, ro = orc.GMG(H["mats"], H["prolongations"], maxiter=1).smooth(0, np.zeros(n), x)
    assert max_rel(xs, xo) <= TOL_KERNEL and max_rel(rs, ro) <= TOL_KERNEL


def test_drop_in_preconditioner_under_a_host_cg(S, po, orc, hierarchy):
    """Integration mode 1: the reference keeps its own CGSolver and only `solve!(z,Pl,r)`
    (CGSolvers.jl:94) goes to the device.  A host-side CG written exactly like CGSolvers.jl:73-120, calling
    the device V-cycle through the host-memory ABI, must reproduce the oracle's iteration."""
    nc, nlev = (16, 16, 16), 3
    H = hierarchy(nc, nlev)
    A = H["mats"][0].to_scipy()
    b = po.dirichlet_lift_rhs(nc, 1)
    gmg = make_gmg(S, H)                                   # maxiter=1, :preconditioner
    Pl = setup(S, gmg, H["mats"][0])
    x = np.zeros_like(b); r = b - A @ x; p = np.zeros_like(b); z = np.zeros_like(b)
    gamma = 1.0
    res0 = res = np.linalg.norm(r); hist = [res]
    it = 0
    while not (it >= 20 or (it > 0 and res / res0 < 1e-6) or res < 1e-14):
        S.solve_(z, Pl, r)                                  # device V-cycle, host vectors
        beta = gamma; gamma = float(z @ r); beta = gamma / beta
        p = z + beta * p
        w = A @ p
        alpha = gamma / float(p @ w)
        x += alpha * p; r -= alpha * w
        res = np.linalg.norm(r); hist.append(res); it += 1
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=1)
    xo, nit, flag, histo = orc.cg_solve(H["mats"][0], b, Pl=go, maxiter=20, atol=1e-14, rtol=1e-6)
    assert it == nit
    np.testing.assert_allclose(hist, histo, rtol=TOL_HIST)
    assert rel_err(x, xo) <= 1e-10
    assert gmg.log.num_iters == 1 and gmg.log.residuals[0] > gmg.log.residuals[1] > 0      # GMG's own ConvergenceLog is filled


@pytest.mark.parametrize("nc,order,pivot", [((8, 8, 8), 2, True), ((16, 16), 2, False), ((16, 16, 16), 1, True)])
def test_patch_corrected_prolongation_parity(S, po, orc, hierarchy, nc, order, pivot):
    """GMG with PatchProlongationOperator transfers (test/LinearSolvers/GMGTests.jl:80-88 ttype=:patch)
    against the oracle: V-cycle output and an FGMRES solve."""
    nlev = 2
    H = hierarchy(nc, nlev, order)
    coarse = tuple(c // 2 for c in nc)
    pp, pd = po.coarse_cell_interior_patches(coarse, order)
    interp = [S.PatchProlongationOperator(H["prolongations"][0], pp, pd, pivoting=pivot)]
    Hp = dict(mats=H["mats"], prolongations=interp, restrictions=H["restrictions"])
    gmg = make_gmg(S, Hp)
    solver = S.FGMRESSolver(5, gmg, maxiter=20, atol=1e-14, rtol=1e-8)
    ns = setup(S, solver, H["mats"][0])
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=1,
                 prolongation_patches=[(orc.PATCH if pivot else orc.BLOCKJACOBI, pp, pd)])
    r = seeded(H["mats"][0].shape[0], 29)
    z = np.zeros_like(r)
    S.solve_(z, ns.P_ns, r)
    zo, _, _, ho = go.solve(r)
    assert rel_err(z, zo) <= TOL_VCYCLE
    np.testing.assert_allclose(gmg.log.residuals[:2], ho, rtol=TOL_HIST)
    b = po.dirichlet_lift_rhs(nc, order)
    x = np.zeros_like(b)
    S.solve_(x, ns, b)
    xo, nit, flag, hist = orc.fgmres_solve(H["mats"][0], b, Pr=go, m=5, maxiter=20, atol=1e-14, rtol=1e-8)
    assert solver.log.num_iters == nit and rel_err(x, xo) <= 1e-9
    assert po.l2_error_sq(nc, order, x) < 1e-8


def test_device_coarse_inversion_matches_host_factorisation(S, po, orc, hierarchy, monkeypatch):
    """Large coarsest levels are inverted on the device (blocked Gauss-Jordan, verified at setup); forced here
    on a 3375-dof coarse level and compared with the pivoted banded LU of the host path and with the oracle."""
    nc, nlev = (32, 32, 32), 2                                 # coarsest = 16^3 cells -> 3375 dofs
    H = hierarchy(nc, nlev)
    rc = seeded(H["mats"][-1].shape[0], 31)
    out = {}
    for name, lim in (("host", "100000"), ("device", "0")):
        monkeypatch.setenv("GMG_COARSE_HOST_MAX", lim)
        ns = setup(S, make_gmg(S, H), H["mats"][0])
        x = np.zeros_like(rc)
        ns.coarse_solve(rc, x)
        out[name] = x
    xo = orc.GMG(H["mats"], H["prolongations"], maxiter=1).coarse_solve(rc)
    assert max_rel(out["host"], xo) <= 1e-12 and max_rel(out["device"], xo) <= 1e-11
    assert max_rel(out["device"], out["host"]) <= 1e-11


def test_row_pattern_format_details(S, po, orc, hierarchy, monkeypatch):
    """SELL-P (row-pattern dictionary): square operators use offsets relative to the row (shared-offset kernel with
    DPP neighbour exchange), the rectangular transfers offsets relative to their first column; domain ends are clamped;
    a non-finite entry of x must only reach the rows that hold a stored coefficient for it; matrices whose rows are
    all different fall back to the other layouts."""
    from gridapsolvers_jl_amd import abi
    nc, nlev = (12, 20, 8), 2                                  # 11*19*7 = 1463 rows: ragged last slice, many edge slices
    H = hierarchy(nc, nlev)
    ns = setup(S, make_gmg(S, H), H["mats"][0])
    assert ns.level_format(0)["layout"] == "SELL-P" and ns.level_format(0)["stream_bytes_per_nnz"] < 0.2
    A, P, R = H["mats"][0], H["prolongations"][0], H["restrictions"][0]
    x = seeded(A.shape[0], 5)
    for op, M, xin in ((abi.OP_A, A, x), (abi.OP_R, R, x), (abi.OP_P, P, seeded(P.shape[1], 6))):
        y = np.zeros(M.shape[0])
        ns.op_apply(0, op, xin, y)
        assert np.array_equal(y, orc.spmv(M, xin)), op         # sequential row sums in CSR order: bit-identical
    # an Inf in x reaches exactly the rows that store a coefficient for that column
    k = A.shape[0] // 2
    xi = x.copy(); xi[k] = np.inf
    y = np.zeros_like(xi)
    ns.op_apply(0, abi.OP_A, xi, y)
    touched = np.zeros(A.shape[0], dtype=bool)
    As = A.to_scipy().tocsc()
    touched[As.indices[As.indptr[k]:As.indptr[k + 1]]] = True
    assert np.all(np.isfinite(y[~touched])) and not np.any(np.isfinite(y[touched]))
    assert np.array_equal(y[~touched], orc.spmv(A, x)[~touched])
    ns.close()
    # all rows different (random symmetric scaling): no pattern dictionary
    D = np.random.default_rng(8).uniform(0.5, 2.0, A.shape[0])
    As = A.to_scipy().multiply(D[:, None]).multiply(D[None, :]).tocsr(); As.sort_indices()
    A2 = po.CSR(A.shape, As.indptr, As.indices, As.data)
    H2 = dict(mats=[A2, H["mats"][1]], prolongations=H["prolongations"], restrictions=H["restrictions"])
    ns2 = setup(S, make_gmg(S, H2), A2)
    assert ns2.level_format(0)["layout"] != "SELL-P"
    y = np.zeros_like(x)
    ns2.op_apply(0, abi.OP_A, x, y)
    assert max_rel(y, orc.spmv(A2, x)) <= 1e-15
    ns2.close()


def test_row_pattern_unsorted_and_duplicate_columns(S, po, orc, pkg):
    """Rows given with unsorted or repeated column indices keep their summation order: the shared-offset form
    (ascending offsets) must not be chosen for them, the generic pattern form sums in the given order."""
    from gridapsolvers_jl_amd import abi
    n = 256
    ptr = np.arange(0, 4 * n + 1, 4, dtype=np.int64)
    idx = np.zeros(4 * n, dtype=np.int32)
    val = np.zeros(4 * n)
    for i in range(n):
        cols = [(i + 1) % n, i, (i + n - 1) % n, i]             # unsorted, diagonal twice
        idx[4 * i:4 * i + 4] = cols
        val[4 * i:4 * i + 4] = [-1.0, 2.5, -1.0, 1.5]
    A = po.CSR((n, n), ptr, idx, val)
    Ac = po.CSR((2, 2), [0, 1, 2], [0, 1], [1.0, 1.0])
    Pm = po.CSR((n, 2), np.arange(n + 1), (np.arange(n) % 2).astype(np.int32), np.ones(n))
    gmg = S.GMGLinearSolver([A, Ac], [Pm], None, pre_smoothers=jac(S, 2, 1, 0.5), post_smoothers=jac(S, 2, 1, 0.5), maxiter=1)
    ns = setup(S, gmg, A)
    x = seeded(n, 2)
    y = np.zeros(n)
    ns.op_apply(0, abi.OP_A, x, y)
    assert np.array_equal(y, orc.spmv(A, x))
    ns.close()


@pytest.mark.parametrize("nc", [(8, 8), (8, 8, 8), (32, 32, 32)])
def test_reference_smoothers_test_on_device(S, po, orc, hierarchy, nc):
    """test/LinearSolvers/SmoothersTests.jl:13-43,58-74 on the device: CGSolver(LinearSolverFromSmoother(
    RichardsonSmoother(JacobiLinearSolver(),5,2/3)); rtol=1e-8), u = x1 + x2, `@test E < 1.e-8` -- the reference's own
    known-answer criterion -- plus parity with the oracle's literal execution of the same solver."""
    H = hierarchy(nc, 2)
    sm = S.RichardsonSmoother(S.JacobiLinearSolver(), 5, 2.0 / 3.0)
    gmg = S.GMGLinearSolver(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=[sm], post_smoothers=[sm], maxiter=1)
    solver = S.CGSolver((S.LinearSolverFromSmoother(sm), gmg), maxiter=1000, atol=1e-12, rtol=1e-8)
    ns = setup(S, solver, H["mats"][0])
    b = po.dirichlet_lift_rhs(nc, 1)
    x = np.zeros_like(b)
    S.solve_(x, ns, b)
    assert solver.log.flag in (S.SOLVER_CONVERGED_ATOL, S.SOLVER_CONVERGED_RTOL)
    assert po.l2_error_sq(nc, 1, x) < 1.0e-8                              # SmoothersTests.jl:43
    xo, nit, flag, hist = orc.cg_smoother_solve(H["mats"][0], b, 5, 2.0 / 3.0, maxiter=1000, atol=1e-12, rtol=1e-8)
    assert solver.log.num_iters == nit
    np.testing.assert_allclose(solver.log.residuals[: nit + 1], hist, rtol=1e-6, atol=1e-9 * hist[0])
    assert rel_err(x, xo) <= 1e-8


@pytest.mark.parametrize("nc", [(8, 8), (8, 8, 8)])
@pytest.mark.parametrize("pl", ["jacobi", "none", "gmg"])
def test_reference_richardson_linear_test_on_device(S, po, orc, hierarchy, nc, pl):
    """test/LinearSolvers/RichardsonLinearTests.jl:14-26,64-73 on the device (`@test E < 1.e-6`), parity with the oracle;
    Pl = GMG additionally (the Richardson iteration as the outer caller of the V-cycle)."""
    H = hierarchy(nc, 2)
    gmg = make_gmg(S, H)
    P = {"jacobi": (S.JacobiLinearSolver(), gmg), "none": (None, gmg), "gmg": gmg}[pl]
    solver = S.RichardsonLinearSolver(0.5, 1000, Pl=P, rtol=1e-8)
    ns = setup(S, solver, H["mats"][0])
    b = po.dirichlet_lift_rhs(nc, 1)
    x = np.zeros_like(b)
    S.solve_(x, ns, b)
    assert po.l2_error_sq(nc, 1, x) < 1.0e-6                              # RichardsonLinearTests.jl:26
    Po = {"jacobi": "jacobi", "none": None, "gmg": orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=1)}[pl]
    xo, nit, flag, hist = orc.richardson_solve(H["mats"][0], b, 0.5, Pl=Po, maxiter=1000, rtol=1e-8)
    assert solver.log.num_iters == nit and solver.log.flag == flag
    np.testing.assert_allclose(solver.log.residuals[: nit + 1], hist, rtol=1e-6, atol=1e-9 * hist[0])
    assert rel_err(x, xo) <= 1e-8


@pytest.mark.parametrize("seed,n,noff", [(1, 1000, 5), (2, 4099, 11), (3, 700, 2), (4, 2500, 30)])
def test_row_pattern_random_banded_and_circulant(S, po, orc, seed, n, noff):
    """Row-pattern layouts on matrices that are NOT finite-element stencils: a circulant matrix with random offsets
    (wrap-around rows have their own patterns; runs of 1-3 offsets, dummy runs padded in), its banded truncation, and a
    rectangular restriction-like operator (offsets relative to the first column).  Bit-identical to the sequential SpMV."""
    from gridapsolvers_jl_amd import abi
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    offs = np.unique(np.concatenate([[0], rng.integers(-n // 3, n // 3, size=noff)]))
    vals = rng.uniform(-1.0, 1.0, offs.size); vals[offs == 0] = 4.0 + offs.size
    rows = np.repeat(np.arange(n), offs.size)
    cols = (rows + np.tile(offs, n)) % n
    Acirc = sp.csr_matrix((np.tile(vals, n), (rows, cols)), shape=(n, n)); Acirc.sum_duplicates(); Acirc.sort_indices()
    keep = np.abs((rows + np.tile(offs, n)) - cols) == 0            # entries that did not wrap
    Aband = sp.csr_matrix((np.tile(vals, n)[keep], (rows[keep], cols[keep])), shape=(n, n)); Aband.sort_indices()
    nc = n // 2
    prow = np.repeat(np.arange(n), 2); pcol = np.minimum(np.stack([np.arange(n) // 2, np.arange(n) // 2 + 1], 1).ravel(), nc - 1)
    Pm = sp.csr_matrix((np.tile([0.75, 0.25], n), (prow, pcol)), shape=(n, nc)); Pm.sum_duplicates(); Pm.sort_indices()
    Ac = sp.identity(nc, format="csr") * 3.0
    for A in (Acirc, Aband):
        H = dict(mats=[po.CSR(A.shape, A.indptr, A.indices, A.data), po.CSR(Ac.shape, Ac.indptr, Ac.indices, Ac.data)],
                 prolongations=[po.CSR(Pm.shape, Pm.indptr, Pm.indices, Pm.data)], restrictions=[None])
        gmg = S.GMGLinearSolver(H["mats"], H["prolongations"], None, pre_smoothers=jac(S, 2, 2, 0.5), post_smoothers=jac(S, 2, 2, 0.5), maxiter=1)
        ns = setup(S, gmg, H["mats"][0])
        assert ns.level_format(0)["layout"] == "SELL-P"
        x = seeded(n, 10 + seed)
        y = np.zeros(n); ns.op_apply(0, abi.OP_A, x, y)
        assert np.array_equal(y, orc.spmv(H["mats"][0], x))
        xc = seeded(nc, 20 + seed)
        yp = np.zeros(n); ns.op_apply(0, abi.OP_P, xc, yp)
        assert np.array_equal(yp, orc.spmv(H["prolongations"][0], xc))
        yr = np.zeros(nc); ns.op_apply(0, abi.OP_R, x, yr)
        Rt = Pm.T.tocsr(); Rt.sort_indices()
        assert max_rel(yr, orc.spmv(po.CSR(Rt.shape, Rt.indptr, Rt.indices, Rt.data), x)) <= 1e-15
        # a smoothing pass (fused sweeps with deferred x update, table D^-1) against the oracle's literal loop
        go = orc.GMG(H["mats"], H["prolongations"], pre_smoothers=[orc.Smoother(orc.JACOBI, 2, 0.5)], maxiter=1)
        x0, r0 = seeded(n, 30 + seed), seeded(n, 40 + seed)
        xs, rs = x0.copy(), r0.copy()
        ns.smooth(0, xs, rs)
        xo, ro = go.smooth(0, x0, r0)
        assert max_rel(xs, xo) <= TOL_KERNEL and max_rel(rs, ro) <= TOL_KERNEL
        ns.close()


def test_row_pattern_coded_table_q2(S, po, orc, hierarchy, monkeypatch):
    """Q2 stiffness matrices have 216 row patterns of up to 125 entries: too wide for the 12-byte-per-entry tables, so the
    shared-offset kernel runs on a coded table (one byte per entry into a dictionary of the 38 distinct values, runs of
    5 consecutive offsets).  Forced on a small problem here (by default only levels >= 5e5 rows take it)."""
    from gridapsolvers_jl_amd import abi
    monkeypatch.setenv("GMG_PAT_CODED_MIN_ROWS", "0")
    nc, nlev = (8, 8, 8), 2
    H = hierarchy(nc, nlev, 2)
    pp, pd = po.vertex_star_patches(nc, 2)
    sm = [S.RichardsonSmoother(S.PatchSolver(pp, pd), 3, 0.2)]
    gmg = S.GMGLinearSolver(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=sm, post_smoothers=sm, maxiter=1)
    ns = setup(S, gmg, H["mats"][0])
    fmt = ns.level_format(0)
    assert fmt["layout"] == "SELL-P" and fmt["stream_bytes_per_nnz"] < 0.1
    A = H["mats"][0]
    x = seeded(A.shape[0], 77)
    y = np.zeros_like(x)
    ns.op_apply(0, abi.OP_A, x, y)
    assert np.array_equal(y, orc.spmv(A, x))                  # explicit zeros of the superset add nothing: bit-identical
    r = seeded(A.shape[0], 78)
    z = np.zeros_like(r)
    S.solve_(z, ns, r)
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=[orc.Smoother(orc.PATCH, 3, 0.2, pp, pd)], maxiter=1)
    assert rel_err(z, go.solve(r)[0]) <= TOL_VCYCLE
    # Jacobi sweeps on the same operator (fused sweep epilogue of the coded kernel, table D^-1)
    gj = S.GMGLinearSolver(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=jac(S, 2, 3, 0.5), post_smoothers=jac(S, 2, 3, 0.5), maxiter=1)
    nj = setup(S, gj, A)
    x0, r0 = seeded(A.shape[0], 1), seeded(A.shape[0], 2)
    xs, rs = x0.copy(), r0.copy()
    nj.smooth(0, xs, rs)
    xo, ro = orc.GMG(H["mats"], H["prolongations"], pre_smoothers=[orc.Smoother(orc.JACOBI, 3, 0.5)], maxiter=1).smooth(0, x0, r0)
    assert max_rel(xs, xo) <= TOL_KERNEL and max_rel(rs, ro) <= TOL_KERNEL
    ns.close(); nj.close()
