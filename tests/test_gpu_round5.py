"""GPU tests of round 5: the kernels that carry the levels of >= pat_tile_rows rows (sells_r2sweep_kernel<..., OCC=2> with one
slice per wave in eight-wave workgroups, masks in global memory; sells_r2mv_kernel<EPI, ..., OCC=2>) put under the oracle on
small levels (option pat_tile_rows = 0), and the two workloads the bench runs beyond the oracle's reach -- the per-GPU problem
of BASELINE configs[3] (288^3 Q1, 6 levels) and BASELINE configs[2] at its stated size (256^3 Q2, 5 levels) -- through their
size-independent properties.  Same bar as tests/test_gpu_parity.py: the HIP path through the C ABI."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL_KERNEL = 1e-13   # per-kernel max|y - y_ref| / max|y_ref|


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


def max_rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


# the levels: a cube, a box with three different extents, one whose row count leaves a ragged last slice AND a ragged last
# workgroup (39^3 = 59 319 rows = 470 slices of 126 + 99 rows = 58 workgroups of eight slices + 7), a tiny one (one workgroup)
BIG_LEVEL_CASES = [((40, 40, 40), 3, 10), ((34, 46, 30), 2, 3), ((26, 22, 58), 2, 7), ((12, 12, 12), 2, 4)]


# ---------------------------------------------------------------- big-level sweep (OCC = 2) against the single-row sweep and the oracle
BIG_FORMS = {"occ2": ({"pat_tile_rows": 0, "pat_zwalk": 0, "persist": 0}, ("sells_r2sweep_kernel", "OCC=2", "wpb=8")),
             "zwalk": ({"pat_zwalk": 2, "persist": 0}, ("sells_zsweep_kernel", "T=12")),
             "zwalk_T3": ({"pat_zwalk": 2, "pat_zwalk_T": 3, "persist": 0}, ("sells_zsweep_kernel", "T=3")),
             "zwalk_T1": ({"pat_zwalk": 2, "pat_zwalk_T": 1, "persist": 0}, ("sells_zsweep_kernel", "T=1"))}


# round 6: two sweeps per launch (sells_z2sweep_kernel; default on grid levels of >= pat_fuse2_rows = 1e6 rows of one GPU), forced onto the
# small levels: tile shapes from the default to the smallest legal one (three lines per workgroup, one plane per block: every row of
# sweep k is somebody's rim), and two levels whose grid lines are longer than a wave holds (135 and 259 nodes: two / three x segments
# with one node of sweep k either side).  fuse2 = the form a constant-coefficient box operator gets (BC: one set of coefficients per wave from the
# kernel arguments, verified row by row at setup), fuse2_general = per-row patterns from LDS (what any other grid operator gets).
# box* = sells_boxsweep_kernel (round 6): the SINGLE sweep of a constant-coefficient box with the coefficients from the kernel arguments (default on
# levels of 1e6 .. 9e6 rows), forced onto the small levels with chains of the default length, of one plane and of five
BIG_FORMS.update({"fuse2": ({"pat_fuse2": 2, "persist": 0}, ("sells_z2sweep_kernel", "BC=1")),
                  "fuse2_general": ({"pat_fuse2": 2, "pat_fuse2_box": 0, "persist": 0}, ("sells_z2sweep_kernel", "BC=0")),
                  "fuse2_W16_T3": ({"pat_fuse2": 2, "pat_fuse2_w": 16, "pat_fuse2_t": 3, "persist": 0}, ("sells_z2sweep_kernel", "T=3")),
                  "fuse2_W3_T1": ({"pat_fuse2": 2, "pat_fuse2_w": 3, "pat_fuse2_t": 1, "persist": 0}, ("sells_z2sweep_kernel", "W=3", "T=1"))})
BIG_FORMS.update({"box": ({"pat_box": 2, "persist": 0}, ("sells_boxsweep_kernel",)),
                  "box_T1": ({"pat_box": 2, "pat_box_t": 1, "persist": 0}, ("sells_boxsweep_kernel", "T=1")),
                  "box_T5": ({"pat_box": 2, "pat_box_t": 5, "persist": 0}, ("sells_boxsweep_kernel", "T=5"))})
LONG_LINE_CASES = [((136, 8, 8), 2, 4), ((260, 8, 16), 2, 10)]
SWEEP_CASES = [(c, f) for f in BIG_FORMS for c in BIG_LEVEL_CASES] + [(c, f) for f in ("fuse2", "fuse2_general", "fuse2_W3_T1", "box", "box_T5") for c in LONG_LINE_CASES]


@pytest.mark.parametrize("case,form", SWEEP_CASES, ids=lambda v: v if isinstance(v, str) else "x".join(map(str, v[0])) + f"-{v[1]}-{v[2]}")
def test_big_level_sweep_is_bitwise_the_single_row_sweep_and_matches_the_oracle(S, po, orc, case, form):
    """sells_r2sweep_kernel<XM, MK, FM, 9, OCC=2> -- the form every Q1 level of >= pat_tile_rows (3.5e6) rows takes: 288^3, 256^3,
    the finest level of BASELINE configs[3] -- forced onto small levels with pat_tile_rows = 0 and compared (i) bit for bit with
    sells_rsweep_kernel (pat_r2 = 0, one row per lane, masks in LDS) and (ii) with the oracle's literal
    RichardsonSmoothers.jl:84-98 loop to 1e-13: passes from a given x and from x = 0, odd and even sweep counts (both xmode
    variants, the deferred x update), chained passes, +-Inf and NaN in r (the slice's vote fails and the redo path reads the
    masks from GLOBAL memory -- the `GM` branch that no small level reaches by default), and a whole CG solve.
    form = zwalk*: the same for sells_zsweep_kernel (round 5: the pair sweep as a walk along the slowest grid direction -- an interval
    of <= 126 rows of a grid plane per wave, three new windows per step, six carried in registers; default on levels of >=
    pat_zwalk_rows = 9e6 rows), forced onto the small levels with pat_zwalk = 2, with chains of 12, 3 and 1 planes (chain starts
    and ends in the middle of the level, planes whose row count is not a multiple of the interval, the clamped first / last planes).
    form = fuse2*: sells_z2sweep_kernel (round 6), sweeps k and k + 1 of a pass in one launch with r_{k+1} in LDS only: the same bits as
    the single sweeps for every tile shape, from x given / x = 0, odd sweep counts (the last sweep runs alone), Inf / NaN in r."""
    (nc, nlev, niter) = case
    big_opts, big_sig = BIG_FORMS[form]
    if form.startswith("fuse2") and niter % 2:
        big_sig = ()                                      # (odd pass: the last sweep -- and with it the signature -- is a single one)
    H = po.build_hierarchy(nc, nlev, 1)
    n = H["mats"][0].shape[0]
    b = po.dirichlet_lift_rhs(nc, 1)
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=[orc.Smoother(orc.JACOBI, niter, 2.0 / 3.0)] * (nlev - 1), maxiter=1)
    x0, r0 = np.random.default_rng(3).uniform(-1, 1, n), np.random.default_rng(50).uniform(-1, 1, n)
    ri0 = np.random.default_rng(51).uniform(-1, 1, n)
    ri0[n // 3] = np.inf
    ri0[5] = -np.inf
    ri0[(2 * n) // 3] = np.nan
    ri0[n - 2] = np.inf                                   # inside the ragged last slice
    res = {}
    for key, opts in (("big", big_opts), ("single", {"pat_r2": 0, "persist": 0})):
        solver = S.CGSolver(make_gmg(S, H, pre_smoothers=jac(S, nlev, niter), options=opts), maxiter=40, atol=1e-14, rtol=1e-8)
        ns = setup(S, solver, H["mats"][0])
        out = []
        x, r = x0.copy(), r0.copy()
        ns.P_ns.smooth(0, x, r)
        out += [x.copy(), r.copy()]                       # one pass: compared with the oracle below
        for _ in range(2):
            ns.P_ns.smooth(0, x, r)
        out += [x.copy(), r.copy()]
        xz, rz = np.zeros(n), r0.copy()
        ns.P_ns.smooth(0, xz, rz)
        out += [xz, rz]
        xi, ri = np.zeros(n), ri0.copy()
        ns.P_ns.smooth(0, xi, ri)
        out += [np.isfinite(xi), np.isfinite(ri), np.isnan(ri), np.where(np.isfinite(xi), xi, 0.0), np.where(np.isfinite(ri), ri, 0.0)]
        xs = np.zeros(n)
        S.solve_(xs, ns, b)
        out += [xs, solver.log.residuals[: solver.log.num_iters + 1].copy()]
        sig = ns.P_ns.sweep_signature(0)
        if key == "big":
            assert all(t in sig for t in big_sig), sig
        else:
            assert "sells_rsweep_kernel" in sig, sig
        res[key] = out
        ns.P_ns.close()
    for a, c in zip(res["big"], res["single"]):
        np.testing.assert_array_equal(a, c)
    B = res["big"]
    # (ii) the oracle: one pass from (x0, r0), one from x = 0, and the pass with non-finite entries
    xo, ro = go.smooth(0, x0, r0)
    assert max_rel(B[0], xo) <= TOL_KERNEL and max_rel(B[1], ro) <= TOL_KERNEL
    xo, ro = go.smooth(0, np.zeros(n), r0)
    assert max_rel(B[4], xo) <= TOL_KERNEL and max_rel(B[5], ro) <= TOL_KERNEL
    with np.errstate(all="ignore"):
        xo, ro = go.smooth(0, np.zeros(n), ri0)
    np.testing.assert_array_equal(B[6], np.isfinite(xo))              # non-finite values reach exactly the rows they reach in the reference
    np.testing.assert_array_equal(B[7], np.isfinite(ro))
    fin = B[6] & B[7]
    assert 0 < (~fin).sum() < n                                          # (on the small levels the values spread far; never everywhere, never nowhere)
    if fin.sum() > 0:
        assert max_rel(B[9][fin], xo[fin]) <= TOL_KERNEL and max_rel(B[10][fin], ro[fin]) <= TOL_KERNEL
    xo, nit, flag_o, hist = orc.cg_solve(H["mats"][0], b, Pl=go, maxiter=40, atol=1e-14, rtol=1e-8)
    assert len(B[-1]) == nit + 1
    if nit < 40:                                                         # converged: the whole history and the solution
        np.testing.assert_allclose(B[-1], hist, rtol=1e-8)
        assert rel_err(B[-2], xo) <= 1e-10
    else:                                                                # (stretched cells: CG runs into maxiter and amplifies rounding differences late)
        np.testing.assert_allclose(B[-1][:13], hist[:13], rtol=1e-8)
        assert rel_err(B[-2], xo) <= 1e-3


# ---------------------------------------------------------------- big-level mat-vecs (OCC = 2) against the single-row kernels and the oracle
MV_FORMS = {"occ2": {"pat_tile_rows": 0, "pat_zwalk": 0, "pat_r2mv_min": 1, "pat_r2mv_dot": 0, "persist": 0},
            "zwalk": {"pat_zwalk": 2, "pat_r2mv_min": 1, "pat_r2mv_dot": 0, "persist": 0},
            "zwalk_T2": {"pat_zwalk": 2, "pat_zwalk_T": 2, "pat_r2mv_min": 1, "pat_r2mv_dot": 0, "persist": 0}}


@pytest.mark.parametrize("form", list(MV_FORMS))
@pytest.mark.parametrize("nc,nlev,niter", BIG_LEVEL_CASES)
def test_big_level_matvecs_are_bitwise_the_single_row_kernels_and_the_oracle(S, po, orc, nc, nlev, niter, form):
    """sells_r2mv_kernel<EPI_SET / EPI_SUB / EPI_RESID, MK, FM, 9, false, OCC=2> (w = A p of CGSolvers.jl:104 without the fused dot,
    r -= A dx of GMGLinearSolvers.jl:495, r = b - A x of CGSolvers.jl:79 on levels of >= pat_tile_rows rows), forced with
    pat_tile_rows = 0, against sells_kernel (pat_r2mv = 0): y = A x equals the oracle's mul! bit for bit (rows summed in CSR
    order), also with +-Inf / NaN in x (confined to the rows that store a coefficient for them); a CG solve from a random guess
    (RESID at the start, SET every iteration, SUB in every V-cycle) gives the same bits as the single-row kernels and the oracle's
    iteration count and history.  form = zwalk*: the same three mat-vecs in the z-walk form (sells_zsweep_kernel<1, MK, FM, EPI>, the
    default on levels of >= pat_zwalk_rows rows since round 5), sweeps included."""
    from gridapsolvers_jl_amd.abi import OP_A
    H = po.build_hierarchy(nc, nlev, 1)
    A = H["mats"][0]
    n = A.shape[0]
    b = po.dirichlet_lift_rhs(nc, 1)
    xr = np.random.default_rng(7).uniform(-1, 1, n)
    xi = xr.copy()
    xi[n // 2] = np.inf
    xi[3] = -np.inf
    xi[n - 1] = np.nan
    guess = np.random.default_rng(8).uniform(-1, 1, n)
    res = {}
    for key, opts in (("big", MV_FORMS[form]), ("single", {"pat_r2mv": 0, "pat_r2": 0, "persist": 0})):
        solver = S.CGSolver(make_gmg(S, H, pre_smoothers=jac(S, nlev, niter), options=opts), maxiter=40, atol=1e-14, rtol=1e-8)
        ns = setup(S, solver, A)
        y = np.zeros(n)
        ns.P_ns.op_apply(0, OP_A, xr, y)
        out = [y.copy()]
        ns.P_ns.op_apply(0, OP_A, xi, y)
        out += [np.isfinite(y), np.isnan(y), np.where(np.isfinite(y), y, 0.0)]
        xs = guess.copy()
        S.solve_(xs, ns, b)
        out += [xs, solver.log.residuals[: solver.log.num_iters + 1].copy()]
        res[key] = out
        ns.P_ns.close()
    for a, c in zip(res["big"], res["single"]):
        np.testing.assert_array_equal(a, c)
    B = res["big"]
    np.testing.assert_array_equal(B[0], orc.spmv(A, xr))
    with np.errstate(all="ignore"):
        yo = orc.spmv(A, xi)
    np.testing.assert_array_equal(B[1], np.isfinite(yo))
    np.testing.assert_array_equal(B[3][B[1]], yo[B[1]])
    assert 0 < (~B[1]).sum() <= 3 * 27
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=[orc.Smoother(orc.JACOBI, niter, 2.0 / 3.0)] * (nlev - 1), maxiter=1)
    xo, nit, flag_o, hist = orc.cg_solve(A, b, x0=guess, Pl=go, maxiter=40, atol=1e-14, rtol=1e-8)
    assert len(B[-1]) == nit + 1
    if nit < 40:                                                         # converged: the whole history and the solution
        np.testing.assert_allclose(B[-1], hist, rtol=1e-8)
        assert rel_err(B[-2], xo) <= 1e-10
    else:                                                                # (stretched cells: CG runs into maxiter and amplifies rounding differences late)
        np.testing.assert_allclose(B[-1][:13], hist[:13], rtol=1e-8)
        assert rel_err(B[-2], xo) <= 1e-3


# ---------------------------------------------------------------- the per-GPU problem of BASELINE configs[3] on its own workload
@pytest.mark.child_process
def test_weak_anchor_288cubed_properties(S, po):
    """288^3 cells Q1, 6 levels (2.37e7 dofs): what every GPU of BASELINE configs[3] (576^3 on 2x2x2) holds and what bench.py reports as
    `weak_anchor_value`.  Beyond the oracle's reach in test time, so: the finest level runs the OCC=2 kernels the tests above pin;
    the finest level runs the big-level kernels (sells_zsweep_kernel, or sells_r2sweep_kernel<OCC=2> with pat_zwalk = 0); CG takes 3 iterations (as the oracle does at every size it reaches with this rhs: 16^3 ... 64^3) with flag = rtol; the
    reference's own L2 criterion (< 1e-8, GMGTests.jl / SmoothersTests.jl:43); the true residual through the device operator;
    and the default (row-pattern) layout reproduces the generic 12 B/nnz layout BIT FOR BIT (every layout sums a row in CSR
    order) -- a checksum over 2.4e7 entries and the residual history, both exact."""
    import torch
    from gridapsolvers_jl_amd import abi
    nc, nlev = (288, 288, 288), 6
    H = po.build_hierarchy(nc, nlev, 1)
    b = po.dirichlet_lift_rhs(nc, 1)
    n = b.size
    bd = torch.from_numpy(b).cuda()
    got = {}
    for key, opts in (("default", None), ("generic", {"vdict": 0, "idx16": 0, "pattern": 0, "opattern": 0})):
        solver = S.CGSolver(make_gmg(S, H, options=opts), maxiter=20, atol=1e-14, rtol=1e-6)
        ns = setup(S, solver, H["mats"][0])
        xd = torch.zeros_like(bd)
        torch.cuda.synchronize()
        S.solve_(xd, ns, bd)
        torch.cuda.synchronize()
        assert solver.log.num_iters == 3 and solver.log.flag == abi.CONVERGED_RTOL, (key, solver.log.num_iters, solver.log.flag)
        hist = np.asarray(solver.log.residuals[:4]).copy()
        assert hist[-1] < 1e-6 * hist[0]
        yd = torch.zeros_like(bd)
        ns.P_ns.op_apply(0, abi.OP_A, xd, yd)
        true_rel = float(torch.linalg.vector_norm(bd - yd) / torch.linalg.vector_norm(bd))
        assert abs(true_rel - hist[-1] / hist[0]) <= 1e-3 * true_rel, (true_rel, hist)       # the recurrence residual IS the true one
        fmt = ns.P_ns.level_format(0)
        sig = ns.P_ns.sweep_signature(0)
        if key == "default":
            assert fmt["layout"] == "SELL-P" and ("sells_zsweep_kernel" in sig or ("sells_r2sweep_kernel" in sig and "OCC=2" in sig)), (fmt, sig)
            assert ns.P_ns.device_bytes() < 4e9
        else:
            assert fmt["layout"] == "SELL-64", fmt
        got[key] = (xd.cpu().numpy(), hist)
        ns.P_ns.close()
        del xd, yd
    x = got["default"][0]
    assert po.l2_error_sq(nc, 1, x) < 1e-8
    assert np.max(np.abs(x - po.nodal_values(nc, 1))) < 1e-5
    np.testing.assert_array_equal(got["default"][1], got["generic"][1])
    assert np.array_equal(got["default"][0], got["generic"][0])


# ---------------------------------------------------------------- BASELINE configs[2] at its stated size
@pytest.mark.child_process
def test_config3_q2_256cubed_properties(S, po):
    """BASELINE configs[2] as stated: 3-D Poisson Q2 on 256^3 cells (1.33e8 dofs, 8.5e9 stored nonzeros, 1.7e7 vertex-star patches on
    the finest level), 5-level GMG, Richardson(PatchSolver,10,0.2) pre = post, FGMRES(5), rtol 1e-6
    (test/LinearSolvers/GMGTests.jl:18-47,119-123).  The three finest operators are streamed (nobody holds their CSR: 102 GB).
    Size-independent properties: 4 iterations as at every size the oracle reaches (8^3 ... 32^3, tests/test_gpu_round2.py /
    round3) and at 128^3; every iteration gains more than a factor 20; flag = rtol; the reference's L2 criterion; the true
    residual through the device operator; device memory < 40 GB of the 288."""
    import time
    import torch
    from gridapsolvers_jl_amd import abi
    nc, nlev, order = (256, 256, 256), 5, 2
    H = po.build_hierarchy(nc, nlev, order, stream_min_rows=1000000)
    assert [hasattr(M, "row_blocks") for M in H["mats"]] == [True, True, True, False, False]
    sm = []
    for l in range(nlev - 1):
        pp, pd = po.vertex_star_patches(H["ncells"][l], order)
        sm.append(S.RichardsonSmoother(S.PatchSolver(pp, pd), 10, 0.2))
    b = po.dirichlet_lift_rhs(nc, order)
    gmg = S.GMGLinearSolver(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=sm, post_smoothers=sm, maxiter=1)
    solver = S.FGMRESSolver(5, gmg, maxiter=20, atol=1e-14, rtol=1e-6)
    t0 = time.time()
    ns = S.numerical_setup(S.symbolic_setup(solver, H["mats"][0]), H["mats"][0])
    t_setup = time.time() - t0
    del sm, gmg.pre_smoothers[:], gmg.post_smoothers[:]
    bd = torch.from_numpy(b).cuda()
    xd = torch.zeros_like(bd)
    torch.cuda.synchronize()
    S.solve_(xd, ns, bd)
    torch.cuda.synchronize()
    assert solver.log.num_iters == 4 and solver.log.flag == abi.CONVERGED_RTOL, (solver.log.num_iters, solver.log.flag)
    hist = np.asarray(solver.log.residuals[:5])
    assert np.all(hist[1:] < 0.05 * hist[:-1])
    yd = torch.zeros_like(bd)
    ns.P_ns.op_apply(0, abi.OP_A, xd, yd)
    assert float(torch.linalg.vector_norm(bd - yd) / torch.linalg.vector_norm(bd)) <= 1.01e-6
    assert ns.P_ns.level_format(0)["row_patterns"] and ns.P_ns.device_bytes() < 40e9
    x = xd.cpu().numpy()
    del yd, bd
    assert po.l2_error_sq(nc, order, x) < 1e-8
    assert np.max(np.abs(x - po.nodal_values(nc, order))) < 1e-5
    assert t_setup < 120.0, t_setup
    ns.P_ns.close()


# ---------------------------------------------------------------- the caller's stream may be HIP's null stream
def test_set_stream_with_the_torch_default_stream_orders_with_the_callers_kernels(S, po, orc):
    """ns.set_stream(torch.cuda.current_stream()): torch's default stream is HIP's null stream, hipStream_t 0 -- which gmg_set_stream
    cannot tell from "NULL = back to the handle's own stream".  The wrappers pass GMG_STREAM_LEGACY (hipStreamLegacy) for it: the
    library's work is then ordered with the caller's default-stream kernels and no synchronisation is needed around solve! -- b is
    produced by a long chain of torch kernels right before the call, x consumed right after.  set_stream(None) returns to the
    handle's own stream."""
    import torch
    nc, nlev = (32, 32, 32), 3
    H = po.build_hierarchy(nc, nlev, 1)
    b = po.dirichlet_lift_rhs(nc, 1)
    n = b.size
    solver = S.CGSolver(make_gmg(S, H), maxiter=20, atol=1e-14, rtol=1e-6)
    ns = setup(S, solver, H["mats"][0])
    own = ns.P_ns.get_stream()
    assert own not in (0, 1, 2)
    assert torch.cuda.current_stream().cuda_stream == 0
    ns.P_ns.set_stream(torch.cuda.current_stream())
    assert ns.P_ns.get_stream() == 1                                   # hipStreamLegacy, not "reset"
    bd0 = torch.from_numpy(b).cuda()
    big = torch.randn(1 << 24, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    res = []
    for _ in range(3):
        # b = b0 + (a long chain on the default stream that sums to exactly 0): unfinished when solve! is called
        acc = torch.zeros((), dtype=torch.float64, device="cuda")
        for _k in range(40):
            acc = acc + (big * 0.0).sum()
        bd = bd0 + acc
        xd = torch.zeros(n, dtype=torch.float64, device="cuda")
        S.solve_(xd, ns, bd)
        y = xd * 1.0                                                   # consumer on the default stream, no synchronize in between
        res.append(y.cpu().numpy())
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=1)
    xo, nit, flag, hist = orc.cg_solve(H["mats"][0], b, Pl=go, maxiter=20, atol=1e-14, rtol=1e-6)
    assert solver.log.num_iters == nit
    for x in res:
        assert rel_err(x, xo) <= 1e-10
        np.testing.assert_array_equal(x, res[0])
    ns.P_ns.set_stream(None)
    assert ns.P_ns.get_stream() == own
    ns.P_ns.close()


# ---------------------------------------------------------------- numerical_setup! on every level, values in CSC order
def test_all_level_refresh_with_csc_values_is_bitwise_a_fresh_setup(S, po, orc):
    """What julia/GridapSolversAMD.jl's numerical_setup!(ns, A, smatrices) does for SparseMatrixCSC inputs (Gridap's default):
    gmg_set_matrix in GMG_CSC layout, then gmg_update_values_csc on EVERY level (GMGLinearSolvers.jl:260-297 recomputes all
    levels, smoothers and the coarsest solver) with the new nzval in CSC order -- the first call hands colptr / rowidx over, later
    calls pass NULL for both -- and gmg_setup.  Variable-coefficient matrices with a skew part (upper triangle scaled by 1 + eps,
    lower by 1 - eps on every level: the values are not symmetric, so CSC order and CSR order really differ): the refreshed setup
    solves bit for bit like a fresh one on the new matrices and like the oracle."""
    import ctypes as C
    import scipy.sparse as sp
    from gridapsolvers_jl_amd import abi
    nc, nlev = (20, 20, 20), 3

    def hierarchy(eps, scale):
        H = po.build_hierarchy(nc, nlev, 1, kappa=po.smooth_kappa)
        mats = []
        for M in H["mats"]:
            A = M.to_scipy().tocsr()
            B = (sp.diags(A.diagonal()) + (1.0 + eps) * sp.triu(A, 1) + (1.0 - eps) * sp.tril(A, -1)) * scale
            B = B.tocsc()
            B.sort_indices()
            assert B.nnz == A.nnz
            mats.append(B)                                             # same pattern, values no longer symmetric
        return H, mats
    H, m1 = hierarchy(0.05, 1.0)
    _H2, m2 = hierarchy(0.11, 1.5)
    b = np.random.default_rng(9).uniform(-1, 1, m1[0].shape[0])

    def solver_for(mats):
        gmg = S.GMGLinearSolver(mats, H["prolongations"], H["restrictions"], pre_smoothers=jac(S, nlev, 4), post_smoothers=jac(S, nlev, 4), maxiter=1)
        return S.FGMRESSolver(10, gmg, maxiter=60, atol=1e-14, rtol=1e-9)
    s_fresh = solver_for(m2)
    nf = setup(S, s_fresh, m2[0])
    xf = np.zeros_like(b)
    S.solve_(xf, nf, b)
    s_ref = solver_for(m1)
    nr = setup(S, s_ref, m1[0])
    x1 = np.zeros_like(b)
    S.solve_(x1, nr, b)
    S.numerical_setup_(nr, m2[0], m2)                                  # every level, CSC values (first call: with colptr / rowidx)
    xr = np.zeros_like(b)
    S.solve_(xr, nr, b)
    assert not np.array_equal(x1, xr)
    np.testing.assert_array_equal(xr, xf)
    np.testing.assert_array_equal(s_ref.log.residuals[: s_ref.log.num_iters + 1], s_fresh.log.residuals[: s_fresh.log.num_iters + 1])
    # back to the first values, this time without the index arrays (the permutation is cached per level)
    lib, h = abi.load(), nr.P_ns.h
    for l, M in enumerate(m1):
        v = np.ascontiguousarray(M.data, dtype=np.float64)
        abi.check(h, lib.gmg_update_values_csc(h, l, None, None, C.c_void_p(v.ctypes.data), 0, 4))
    nr.P_ns.setup()
    xb = np.zeros_like(b)
    S.solve_(xb, nr, b)
    np.testing.assert_array_equal(xb, x1)
    # a level that was not set in CSC layout refuses CSC-ordered values
    csr = [M.tocsr() for M in m1]
    nc_ = setup(S, solver_for(csr), csr[0])
    v = np.ascontiguousarray(m1[0].data)
    st = lib.gmg_update_values_csc(nc_.P_ns.h, 0, C.c_void_p(m1[0].indptr.ctypes.data), C.c_void_p(m1[0].indices.ctypes.data), C.c_void_p(v.ctypes.data), 0, 4)
    assert st == abi.ERR_STATE
    # the oracle on the refreshed matrices
    mo = [po.CSR(M.shape, M.tocsr().indptr, M.tocsr().indices, M.tocsr().data) for M in m2]
    go = orc.GMG(mo, H["prolongations"], H["restrictions"], pre_smoothers=[orc.Smoother(orc.JACOBI, 4, 2.0 / 3.0)] * (nlev - 1), maxiter=1)
    xo, nit, flag, hist = orc.fgmres_solve(mo[0], b, Pr=go, m=10, maxiter=60, atol=1e-14, rtol=1e-9)
    assert s_fresh.log.num_iters == nit
    assert rel_err(xf, xo) <= 1e-9
    for q in (nf, nr, nc_):
        q.P_ns.close()


# ---------------------------------------------------------------- wide rows (Q2) in the z-walk form
@pytest.mark.parametrize("nc,T", [((16, 16, 16), 12), ((24, 10, 12), 3), ((10, 14, 20), 1), ((12, 12, 12), 5)])
def test_wide_row_zwalk_is_bitwise_the_wide_row_kernel_and_the_oracle(S, po, orc, monkeypatch, nc, T):
    """sellw_zwalk_kernel<EPI> (round 5; default on Q2 levels of >= pat_zwalk_rows rows: the finest two levels of BASELINE configs[2]):
    the 25 windows of a row's 5 x 5 runs kept in registers while a wave walks up the grid planes, five new ones gathered per step;
    per-workgroup value tables, runs skipped by plane group.  Forced onto small levels (pat_zwalk = 2, chains of 12 / 3 / 1 / 5
    planes: chain starts and ends inside the level, intervals that straddle grid lines of different dof types, the clamped first and
    last planes).  mul!(y, A, x) equals the oracle's sequential row sums bit for bit; an Inf / NaN in x reaches exactly the rows that
    store a coefficient for it (the step is redone with the coded table's masks); a patch-smoothed FGMRES solve -- r -= A dx
    (EPI_SUB), dx = omega S r ; x += dx (EPI_ADDTO on the additive-Schwarz operator), b - A x (EPI_RESID) -- gives the bits of
    sells_kernel<..., WL> (pat_zwalk = 0) and the oracle's iteration count."""
    from gridapsolvers_jl_amd import abi
    monkeypatch.setenv("GMG_PAT_CODED_MIN_ROWS", "0")               # the coded shared-offset table (what the 10^8-dof levels use) for A and M
    order, nlev = 2, 2
    H = po.build_hierarchy(nc, nlev, order, stream_min_rows=5000)
    Hw = po.build_hierarchy(nc, nlev, order)
    A = Hw["mats"][0]
    n = A.shape[0]
    b = po.dirichlet_lift_rhs(nc, order)
    tabs = [po.vertex_star_patches(c, order) for c in H["ncells"][:-1]]
    x = np.random.default_rng(11).uniform(-1, 1, n)
    ks = [n // 2 + 3, 7, n - 2]
    xi = x.copy(); xi[ks[0]] = np.inf; xi[ks[1]] = -np.inf; xi[ks[2]] = np.nan
    As = A.to_scipy().tocsc()
    touched = np.zeros(n, dtype=bool)
    for k in ks:
        touched[As.indices[As.indptr[k]:As.indptr[k + 1]]] = True
    guess = np.random.default_rng(12).uniform(-1, 1, n)
    sols = {}
    for mode in ("zwalk", "wl"):
        opts = {"pat_zwalk": 2, "pat_zwalk_T": T} if mode == "zwalk" else {"pat_zwalk": 0}
        sm = [S.RichardsonSmoother(S.PatchSolver(pp, pd), 5, 0.2) for pp, pd in tabs]
        solver = S.FGMRESSolver(5, make_gmg(S, H, pre_smoothers=sm, options=opts), maxiter=20, atol=1e-14, rtol=1e-8)
        ns = setup(S, solver, H["mats"][0])
        assert ns.P_ns.level_format(0)["row_patterns"]
        y = np.zeros_like(x)
        ns.P_ns.op_apply(0, abi.OP_A, x, y)
        assert np.array_equal(y, orc.spmv(A, x))
        yi = np.zeros_like(x)
        ns.P_ns.op_apply(0, abi.OP_A, xi, yi)
        assert np.all(np.isfinite(yi[~touched])) and not np.any(np.isfinite(yi[touched]))
        assert np.array_equal(yi[~touched], y[~touched])
        xs = guess.copy()
        S.solve_(xs, ns, b)
        sols[mode] = (xs, solver.log.num_iters, np.array(solver.log.residuals[: solver.log.num_iters + 1]), yi.copy())
        ns.P_ns.close()
    assert sols["zwalk"][1] == sols["wl"][1] and np.array_equal(sols["zwalk"][2], sols["wl"][2]) and np.array_equal(sols["zwalk"][0], sols["wl"][0])
    np.testing.assert_array_equal(np.isnan(sols["zwalk"][3]), np.isnan(sols["wl"][3]))
    osm = [orc.Smoother(orc.PATCH, 5, 0.2, pp, pd) for pp, pd in tabs]
    go = orc.GMG(Hw["mats"], Hw["prolongations"], Hw["restrictions"], pre_smoothers=osm, maxiter=1)
    xo, nit, flag, hist = orc.fgmres_solve(A, b, x0=guess, Pr=go, m=5, maxiter=20, atol=1e-14, rtol=1e-8)
    assert sols["zwalk"][1] == nit
    np.testing.assert_allclose(sols["zwalk"][2], hist, rtol=1e-7)
    assert rel_err(sols["zwalk"][0], xo) <= 1e-9
