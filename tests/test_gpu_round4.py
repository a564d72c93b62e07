"""GPU tests of the round-4 additions: per-handle options (gmg_set_option), host vectors registered once (gmg_host_register),
the x0_zero option, the joint retry of a timed-out one-launch smoothing pass, fused multiply-add taps.  Same bar as
tests/test_gpu_parity.py: the HIP path through the C ABI against the CPU oracle on the same inputs."""
import os

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


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


def cg(S, H, **kw):
    return S.CGSolver(make_gmg(S, H, **kw), maxiter=20, atol=1e-14, rtol=1e-6)


# ---------------------------------------------------------------- layout policy per handle, not per process
def test_options_choose_the_layout_per_handle(S, po, orc):
    """Two solvers in ONE process with different storage layouts (the reference configures by constructor keywords only,
    GMGLinearSolvers.jl:48-58): options pattern=0,vdict=0,idx16=0,opattern=0 give the plain 12 B/nnz SELL-64 stream, the default
    handle keeps the row-pattern form; both reproduce the oracle (iterations identical, histories <= 1e-8) and each other bit for
    bit (every layout sums a row in CSR order).  Unknown keys are rejected; an option set after the setup invalidates it."""
    nc, nlev = (24, 24, 24), 3
    H = po.build_hierarchy(nc, nlev, 1)
    b = po.dirichlet_lift_rhs(nc, 1)
    s1 = cg(S, H)
    s2 = cg(S, H, options={"pattern": 0, "GMG_VDICT": 0, "IDX16": 0, "opattern": 0})
    n1, n2 = setup(S, s1, H["mats"][0]), setup(S, s2, H["mats"][0])
    assert n1.P_ns.level_format(0)["layout"] == "SELL-P" and n2.P_ns.level_format(0)["layout"] == "SELL-64"
    assert n2.P_ns.get_option("pattern") == (0.0, "handle") and n1.P_ns.get_option("pattern") == (None, "default")
    x1, x2 = np.zeros_like(b), np.zeros_like(b)
    S.solve_(x1, n1, b)
    S.solve_(x2, n2, b)
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=1)
    xo, nit, flag, hist = orc.cg_solve(H["mats"][0], b, Pl=go, maxiter=20, atol=1e-14, rtol=1e-6)
    for s, x in ((s1, x1), (s2, x2)):
        assert s.log.num_iters == nit
        np.testing.assert_allclose(s.log.residuals[: nit + 1], hist, rtol=1e-8)
        assert rel_err(x, xo) <= 1e-10
    assert np.array_equal(x1, x2)
    from gridapsolvers_jl_amd import abi
    with pytest.raises(abi.GmgError) as e:
        n1.P_ns.set_option("no_such_option", 1)
    assert e.value.code == abi.ERR_INVALID
    # a layout option after the setup: the handle asks for a new gmg_setup instead of silently keeping the old layout
    n1.P_ns.set_option("pattern", 0)
    with pytest.raises(abi.GmgError) as e:
        S.solve_(x1, n1, b)
    assert e.value.code == abi.ERR_STATE
    n1.P_ns.setup()
    assert n1.P_ns.level_format(0)["layout"] == "SELL-64"
    x3 = np.zeros_like(b)
    S.solve_(x3, n1, b)
    assert np.array_equal(x3, x2)


def test_environment_overrides_the_handle_option(S, po, monkeypatch):
    """GMG_<KEY> in the environment is the debugging override of the same option"""
    nc, nlev = (16, 16, 16), 3
    H = po.build_hierarchy(nc, nlev, 1)
    monkeypatch.setenv("GMG_PATTERN", "0")
    ns = setup(S, cg(S, H, options={"pattern": 1}), H["mats"][0])
    assert ns.P_ns.get_option("pattern") == (0.0, "environment")
    assert ns.P_ns.level_format(0)["layout"] != "SELL-P"


# ---------------------------------------------------------------- host vectors: the path the Julia binding takes
@pytest.mark.parametrize("krylov", ["cg", "fgmres"])
def test_registered_and_unregistered_host_vectors_give_identical_bits(S, po, orc, krylov):
    """b / x as host arrays through GMG_MEM_HOST (what julia/GridapSolversAMD.jl passes): pageable arrays (chunked staging),
    arrays page-locked once (gmg_host_register, the pattern of ext/GridapPETScExt/PETScCaches.jl:23-36), a vector INSIDE a registered
    range, and device tensors all give the same bits; 24 N bytes cross PCIe per solve, 16 N with x0_zero; a non-zero initial guess is
    honoured by default (CGSolvers.jl:79) and ignored with x0_zero."""
    import torch
    nc, nlev = (40, 40, 40), 3        # 59 319 dofs = 474 552 bytes: several staging chunks at host_chunk_bytes = 65536
    H = po.build_hierarchy(nc, nlev, 1)
    b = po.dirichlet_lift_rhs(nc, 1)
    n = b.size
    if krylov == "cg":
        solver = cg(S, H, options={"host_chunk_bytes": 65536})
    else:
        solver = S.FGMRESSolver(5, make_gmg(S, H, options={"host_chunk_bytes": 65536}), maxiter=20, atol=1e-14, rtol=1e-6)
    ns = setup(S, solver, H["mats"][0])
    g = ns.P_ns
    xd = torch.zeros(n, dtype=torch.float64, device="cuda")
    S.solve_(xd, ns, torch.from_numpy(b).cuda())
    x_dev = xd.cpu().numpy()
    st0 = g.host_io_stats()
    x_pg = np.zeros(n)
    S.solve_(x_pg, ns, b.copy())
    st1 = g.host_io_stats()
    assert st1["bytes_up"] - st0["bytes_up"] == 16 * n and st1["bytes_down"] - st0["bytes_down"] == 8 * n and st1["registered"] == 0
    big = np.zeros(3 * n + 5)                         # x lives inside a registered range, at an odd offset
    b_rg, x_rg = b.copy(), big[n + 3: 2 * n + 3]
    g.register_host(big); g.register_host(b_rg)
    assert g.host_io_stats()["registered"] == 2
    S.solve_(x_rg, ns, b_rg)
    assert np.array_equal(x_pg, x_dev) and np.array_equal(x_rg, x_dev)
    assert big[: n + 3].max() == 0.0 and big[2 * n + 3:].max() == 0.0      # nothing written outside x
    # initial guess: honoured by default, ignored (taken as zero, never uploaded) with x0_zero
    rng = np.random.default_rng(3)
    guess = x_dev + 1e-3 * rng.standard_normal(n)
    x_g = guess.copy()
    S.solve_(x_g, ns, b_rg)
    it_guess = solver.log.num_iters
    g.set_option("x0_zero", 1)
    st2 = g.host_io_stats()
    x_z = guess.copy()
    S.solve_(x_z, ns, b_rg)
    st3 = g.host_io_stats()
    assert st3["bytes_up"] - st2["bytes_up"] == 8 * n
    assert np.array_equal(x_z, x_dev) and not np.array_equal(x_g, x_dev)
    assert it_guess <= solver.log.num_iters
    xd2 = torch.from_numpy(guess).cuda()
    S.solve_(xd2, ns, torch.from_numpy(b).cuda())     # device callers: the same option zeroes x in place
    assert np.array_equal(xd2.cpu().numpy(), x_dev)
    g.set_option("x0_zero", 0)
    # memory somebody else page-locked already (here: torch) registers as foreign: used by DMA as it is, never unregistered by the handle
    tp = torch.zeros(n, dtype=torch.float64).pin_memory()
    x_tp = tp.numpy()
    g.register_host(x_tp)
    S.solve_(x_tp, ns, b_rg)
    assert np.array_equal(x_tp, x_dev)
    g.unregister_host(x_tp)
    tp.copy_(torch.from_numpy(x_dev))                 # still page-locked for its owner
    g.unregister_host(big); g.unregister_host(b_rg)
    assert g.host_io_stats()["registered"] == 0
    from gridapsolvers_jl_amd import abi
    with pytest.raises(abi.GmgError):
        g.unregister_host(big)
    # pin_vectors=True: the numerical setup registers the vectors it is called with, once
    s2 = cg(S, H, pin_vectors=True)
    n2 = setup(S, s2, H["mats"][0])
    xa, ba = np.zeros(n), b.copy()
    S.solve_(xa, n2, ba)
    assert n2.P_ns.host_io_stats()["registered"] == 2
    if krylov == "cg":
        assert np.array_equal(xa, x_dev)
    S.solve_(xa, n2, ba)                              # the same objects again: nothing new is registered
    assert n2.P_ns.host_io_stats()["registered"] == 2
    n2.P_ns.close()
    # the oracle, for the record
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=1)
    if krylov == "cg":
        xo, nit, flag, hist = orc.cg_solve(H["mats"][0], b, Pl=go, maxiter=20, atol=1e-14, rtol=1e-6)
    else:
        xo, nit, flag, hist = orc.fgmres_solve(H["mats"][0], b, Pr=go, m=5, maxiter=20, atol=1e-14, rtol=1e-6)
    assert rel_err(x_dev, xo) <= 1e-10


def test_preconditioner_apply_with_host_vectors(S, po, orc):
    """ldiv!(z, ns, r) with host arrays (one V-cycle per call -- the library under a host-language Krylov loop): registered and
    pageable arrays give the oracle's V-cycle (<= 1e-11) and identical bits; r is untouched"""
    nc, nlev = (24, 24, 24), 3
    H = po.build_hierarchy(nc, nlev, 1)
    gmg = make_gmg(S, H)
    ns = setup(S, gmg, H["mats"][0])
    rng = np.random.default_rng(11)
    r = rng.uniform(-1, 1, H["mats"][0].shape[0])
    r_keep = r.copy()
    z1, z2 = np.full_like(r, 7.0), np.full_like(r, -3.0)      # :preconditioner mode overwrites x with zeros first
    S.solve_(z1, ns, r)
    ns.pin(r, z2)
    S.solve_(z2, ns, r)
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=1)
    zo = go.solve(r.copy())[0]
    assert np.array_equal(z1, z2) and np.array_equal(r, r_keep)
    assert rel_err(z1, zo) <= 1e-11


# ---------------------------------------------------------------- one-launch passes: a time-out re-runs the call, on every entry point
def test_forced_timeout_of_a_one_launch_pass_reruns_the_solve(S, po):
    """persist_force_timeout (test hook): the solve behaves as if a one-launch smoothing pass had timed out -- the initial guess is
    restored, the handle switches to per-sweep launches, the call is re-run and returns the bits of an undisturbed solve"""
    import torch
    nc, nlev = (32, 32, 32), 4
    H = po.build_hierarchy(nc, nlev, 1)
    b = po.dirichlet_lift_rhs(nc, 1)
    n = b.size
    ref = setup(S, cg(S, H), H["mats"][0])
    x_ref = np.zeros(n)
    S.solve_(x_ref, ref, b)
    assert ref.P_ns.persist_retries() == dict(retries=0, persist_active=True)
    ns = setup(S, cg(S, H), H["mats"][0])
    ns.P_ns.set_option("persist_force_timeout", 1)
    xd = torch.zeros(n, dtype=torch.float64, device="cuda")   # in place on a device vector: x is restored before the re-run
    S.solve_(xd, ns, torch.from_numpy(b).cuda())
    assert ns.P_ns.persist_retries() == dict(retries=1, persist_active=False)
    assert np.array_equal(xd.cpu().numpy(), x_ref)
    # gmg_smooth directly: x and r come back as from per-sweep launches
    ns2 = setup(S, cg(S, H), H["mats"][0])
    rng = np.random.default_rng(5)
    n1 = H["mats"][1].shape[0]
    x0, r0 = rng.uniform(-1, 1, n1), rng.uniform(-1, 1, n1)
    xa, ra = x0.copy(), r0.copy()
    ref.P_ns.smooth(1, xa, ra)
    ns2.P_ns.set_option("persist_force_timeout", 1)
    xb, rb = torch.from_numpy(x0).cuda(), torch.from_numpy(r0).cuda()
    ns2.P_ns.smooth(1, xb, rb)
    assert ns2.P_ns.persist_retries()["retries"] == 1
    assert np.array_equal(xb.cpu().numpy(), xa) and np.array_equal(rb.cpu().numpy(), ra)


def test_timeout_on_one_rank_is_acted_on_by_all_ranks(tmp_path):
    """Several ranks (host transport on one GPU, one-launch passes on the overlapping / replicated levels): a time-out on ONE rank
    makes EVERY rank restore x and re-run the solve -- the decision is one all-reduce at the end of the solve, no rank leaves the
    collective sequence in the middle -- and the result is the serial oracle's"""
    from test_distributed import _launch, _check
    env = {"GMG_PERSIST_SHARED": "1", "GMG_TEST_DEPTH": "5", "GMG_TEST_FORCE_TIMEOUT_RANK": "1"}
    v = _launch("gpu", 2, (16, 16, 16), 4, tmp_path, transport="host", rep_from=3, extra_env=env)
    _check(v)
    assert v["persist_retries"] == [1, 1] and v["persist_active"] == [0, 0], v


# ---------------------------------------------------------------- two rows per lane, fused multiply-add taps
@pytest.mark.parametrize("nc,nlev,niter", [((40, 40, 40), 3, 10), ((130, 66), 2, 5), ((34, 46, 30), 2, 3)])
def test_pair_sweep_is_bitwise_the_single_row_sweep(S, po, orc, nc, nlev, niter):
    """sells_r2sweep_kernel (default: a lane owns two rows of a 126-row slice, 16-byte loads / stores, masks only applied when a sum
    comes out non-finite) against sells_rsweep_kernel (option pat_r2 = 0) on levels whose row count is not a multiple of 126 and
    whose first / last slices need the clamped path: smoothing passes from a given x and from x = 0, odd and even sweep counts, a
    vector holding Inf (confined to the rows that store a coefficient for it, as in the oracle's mul!), chained passes and a CG
    solve agree to the last bit; the solve equals the oracle's."""
    H = po.build_hierarchy(nc, nlev, 1)
    n = H["mats"][0].shape[0]
    b = po.dirichlet_lift_rhs(nc, 1)
    res = {}
    for r2 in (0, 1):
        solver = S.CGSolver(make_gmg(S, H, pre_smoothers=jac(S, nlev, niter), options={"pat_r2": r2, "persist": 0}), maxiter=40, atol=1e-14, rtol=1e-8)
        ns = setup(S, solver, H["mats"][0])
        out = []
        x, r = np.random.default_rng(3).uniform(-1, 1, n), np.random.default_rng(50).uniform(-1, 1, n)
        for _ in range(3):
            ns.P_ns.smooth(0, x, r)
        out += [x.copy(), r.copy()]
        xi, ri = np.zeros(n), np.random.default_rng(51).uniform(-1, 1, n)
        ri[n // 3] = np.inf
        ri[5] = -np.inf
        ns.P_ns.smooth(0, xi, ri)
        out += [np.isfinite(xi), np.isfinite(ri), np.where(np.isfinite(xi), xi, 0.0), np.where(np.isfinite(ri), ri, 0.0)]
        xs = np.zeros(n)
        S.solve_(xs, ns, b)
        out += [xs, solver.log.residuals[: solver.log.num_iters + 1].copy()]
        sig = ns.P_ns.sweep_signature(0)
        assert ("sells_r2sweep_kernel" in sig) == (r2 == 1) and ("sells_rsweep_kernel" in sig) == (r2 == 0), sig
        res[r2] = out
        ns.P_ns.close()
    for a, c in zip(res[0], res[1]):
        np.testing.assert_array_equal(a, c)
    assert not res[1][2].all() and res[1][2].sum() > 0.9 * n          # the Inf reached some rows and stayed confined
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=[orc.Smoother(orc.JACOBI, niter, 2.0 / 3.0)] * (nlev - 1), maxiter=1)
    xo, nit, flag_o, hist = orc.cg_solve(H["mats"][0], b, Pl=go, maxiter=40, atol=1e-14, rtol=1e-8)
    assert len(res[1][-1]) == nit + 1
    np.testing.assert_allclose(res[1][-1], hist, rtol=1e-8)
    assert rel_err(res[1][-2], xo) <= 1e-10


@pytest.mark.parametrize("nc,nlev,tile", [((32, 32, 32), 4, 0), ((64, 64), 3, 0), ((40, 36, 28), 2, 2)])
def test_fused_multiply_add_taps_stay_inside_the_parity_gates(S, po, orc, nc, nlev, tile, monkeypatch):
    """Option pat_fma = 1 (one rounding per tap instead of the two of the reference's mul!: RichardsonSmoothers.jl:94 through
    SparseArrays): NOT bit-identical, so it is an opt-in -- but it must stay inside the gates SURVEY 8(c) states for the HIP path:
    a smoothing pass <= 1e-13 of the oracle's (max-norm relative), CG iteration count identical, residual history <= 1e-8 per entry,
    solution <= 1e-10."""
    if tile:
        monkeypatch.setenv("GMG_PAT_TILE_MIN", "0")
    H = po.build_hierarchy(nc, nlev, 1)
    n = H["mats"][0].shape[0]
    b = po.dirichlet_lift_rhs(nc, 1)
    # pat_r2mv_min = 1: the pair mat-vec kernels (FM variants of sells_r2mv_kernel, with the fused dot) on these small levels too
    solver = S.CGSolver(make_gmg(S, H, options={"pat_fma": 1, "persist": 0, "pat_tile": tile, "pat_r2mv_min": 1}), maxiter=20, atol=1e-14, rtol=1e-6)
    ns = setup(S, solver, H["mats"][0])
    assert "FM=1" in (ns.P_ns.sweep_signature(0) or "FM=1")
    from gridapsolvers_jl_amd.abi import OP_A
    v = np.random.default_rng(9).uniform(-1, 1, n)
    y = np.zeros(n)
    ns.P_ns.op_apply(0, OP_A, v, y)
    yo = orc.spmv(H["mats"][0], v)
    assert np.max(np.abs(y - yo)) <= 1e-13 * np.max(np.abs(yo)) and not np.array_equal(y, yo)
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=1)
    x, r = np.random.default_rng(3).uniform(-1, 1, n), np.random.default_rng(50).uniform(-1, 1, n)
    xo_, ro_ = go.smooth(0, x.copy(), r.copy())
    ns.P_ns.smooth(0, x, r)
    assert "FM=1" in ns.P_ns.sweep_signature(0)
    assert np.max(np.abs(x - xo_)) <= 1e-13 * np.max(np.abs(xo_)) and np.max(np.abs(r - ro_)) <= 1e-13 * np.max(np.abs(ro_))
    assert not (np.array_equal(x, xo_) and np.array_equal(r, ro_))         # (it really is another arithmetic)
    xs = np.zeros(n)
    S.solve_(xs, ns, b)
    xo, nit, flag, hist = orc.cg_solve(H["mats"][0], b, Pl=go, maxiter=20, atol=1e-14, rtol=1e-6)
    assert solver.log.num_iters == nit
    np.testing.assert_allclose(solver.log.residuals[: nit + 1], hist, rtol=1e-8)
    assert rel_err(xs, xo) <= 1e-10


# ---------------------------------------------------------------- config 3: setup on the device, coarse solver chosen by size
def q2_patch_solver(S, po, nc, nlev, **opts):
    H = po.build_hierarchy(nc, nlev, 2)
    tabs = [po.vertex_star_patches(c, 2) for c in H["ncells"][:-1]]
    sm = [S.RichardsonSmoother(S.PatchSolver(pp, pd), 10, 0.2) for pp, pd in tabs]
    return H, tabs, S.FGMRESSolver(5, make_gmg(S, H, pre_smoothers=sm, options=opts), maxiter=20, atol=1e-14, rtol=1e-6)


def test_patch_operator_lists_built_on_the_device_equal_the_host_path(S, po, orc):
    """Additive-Schwarz operator in row-pattern form (PatchSolvers.jl:279-300 as ONE mat-vec): the counting sort of the patch slots by
    dof and the row signatures on the device (option patch_op_device, default) against the exact host path: identical pattern ids,
    hence bit-identical solves; and the oracle's iteration count / history / solution."""
    nc, nlev = (16, 16, 16), 3
    res = {}
    for dev in (0, 1):
        H, tabs, solver = q2_patch_solver(S, po, nc, nlev, patch_op_device=dev)
        ns = setup(S, solver, H["mats"][0])
        b = po.dirichlet_lift_rhs(nc, 2)
        x = np.zeros_like(b)
        S.solve_(x, ns, b)
        res[dev] = (x, solver.log.residuals[: solver.log.num_iters + 1].copy(), solver.log.num_iters)
        ns.P_ns.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=[orc.Smoother(orc.PATCH, 10, 0.2, pp, pd) for pp, pd in tabs], maxiter=1)
    xo, nit, flag, hist = orc.fgmres_solve(H["mats"][0], b, Pr=go, m=5, maxiter=20, atol=1e-14, rtol=1e-6)
    assert res[1][2] == nit
    np.testing.assert_allclose(res[1][1], hist, rtol=1e-6)
    assert rel_err(res[1][0], xo) <= 1e-9


def test_coarse_solver_chosen_by_size_keeps_the_iteration_count(S, po, orc):
    """A dense-inverse request (LUSolver(), GMGLinearSolvers.jl:54) on a coarsest level of at least coarse_auto_cg_min dofs is served by
    CGSolver(JacobiLinearSolver()) on the device run to rtol 1e-10 (default threshold 20 000: config 3's 29 791-dof coarsest level;
    lowered here): the outer iteration count and flag are those of the exact coarse solve and of the oracle, the history agrees to
    1e-6, the solution to 1e-8; the coarse solver's own log is available."""
    nc, nlev = (16, 16, 16), 3          # coarsest level: 7^3 = 343 dofs
    out = {}
    for amin in (0, 100):
        H, tabs, solver = q2_patch_solver(S, po, nc, nlev, coarse_auto_cg_min=amin)
        ns = setup(S, solver, H["mats"][0])
        b = po.dirichlet_lift_rhs(nc, 2)
        x = np.zeros_like(b)
        S.solve_(x, ns, b)
        out[amin] = (x, solver.log.residuals[: solver.log.num_iters + 1].copy(), solver.log.num_iters, solver.log.flag)
        if amin:
            cl = ns.P_ns.coarse_log()
            assert 5 < cl["niters"] < 2000 and cl["res"] <= 1.0001e-10 * cl["res0"]
        else:
            from gridapsolvers_jl_amd import abi
            with pytest.raises(abi.GmgError):
                ns.P_ns.coarse_log()                      # the exact path has no iterative coarse solve
        ns.P_ns.close()
    assert out[0][2] == out[100][2] and out[0][3] == out[100][3]
    np.testing.assert_allclose(out[100][1], out[0][1], rtol=1e-6)
    assert rel_err(out[100][0], out[0][0]) <= 1e-8
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=[orc.Smoother(orc.PATCH, 10, 0.2, pp, pd) for pp, pd in tabs], maxiter=1)
    xo, nit, flag, hist = orc.fgmres_solve(H["mats"][0], b, Pr=go, m=5, maxiter=20, atol=1e-14, rtol=1e-6)
    assert out[100][2] == nit


def test_posted_norm_equals_the_copied_norm(S, po):
    """host_poll: the residual norm of every Krylov iteration posted into host-mapped memory and polled (default) against the copy +
    hipStreamSynchronize path -- the same kernels produce the number: identical histories and solutions"""
    nc, nlev = (24, 24, 24), 3
    H = po.build_hierarchy(nc, nlev, 1)
    b = po.dirichlet_lift_rhs(nc, 1)
    res = []
    for poll in (1, 0):
        solver = cg(S, H, options={"host_poll": poll})
        ns = setup(S, solver, H["mats"][0])
        x = np.zeros_like(b)
        S.solve_(x, ns, b)
        res.append((x, solver.log.residuals[: solver.log.num_iters + 1].copy()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize("nc,nlev", [((48, 48, 48), 3), ((130, 66), 2), ((34, 46, 30), 2)])
def test_pair_matvec_and_fused_reductions_are_bitwise_the_separate_forms(S, po, orc, nc, nlev):
    """sells_r2mv_kernel (y = A x, y -= A x, y = b - A x with two rows per lane, default on levels of >= pat_r2mv_min rows) against
    sells_kernel (option pat_r2mv = 0), and the reductions finalised by the producing kernel's last workgroup (norm posted to the host
    from there) against dot_partial_kernel + reduce_final_kernel + post_scalar_kernel (red_fused = 0): the operator applied to a
    vector holding Inf (confined to the rows that store a coefficient for it), a CG solve (RESID at the start, SET every iteration,
    SUB in every V-cycle) with its residual history, and a plain dot agree to the last bit; the solve equals the oracle's."""
    from gridapsolvers_jl_amd.abi import OP_A
    H = po.build_hierarchy(nc, nlev, 1)
    n = H["mats"][0].shape[0]
    b = po.dirichlet_lift_rhs(nc, 1)
    res = {}
    for key in ((1, 1), (0, 1), (1, 0), (0, 0), "dot"):
        # "dot": the default -- dot(p, A p) also formed by the mat-vec kernel (another order of the sum: not bitwise, checked below)
        opts = {"pat_r2mv_min": 1} if key == "dot" else {"pat_r2mv": key[0], "pat_r2mv_min": 1, "red_fused": key[1], "pat_r2mv_dot": 0}
        solver = S.CGSolver(make_gmg(S, H, pre_smoothers=jac(S, nlev, 4), options=opts), maxiter=40, atol=1e-14, rtol=1e-8)
        ns = setup(S, solver, H["mats"][0])
        out = []
        x = np.random.default_rng(7).uniform(-1, 1, n)
        y = np.zeros(n)
        ns.P_ns.op_apply(0, OP_A, x, y)
        out.append(y.copy())
        x[n // 2] = np.inf
        x[3] = -np.inf
        ns.P_ns.op_apply(0, OP_A, x, y)
        out += [np.isfinite(y), np.where(np.isfinite(y), y, 0.0)]
        xs = np.random.default_rng(8).uniform(-1, 1, n)      # a guess: y = b - A x at the start
        S.solve_(xs, ns, b)
        out += [xs, solver.log.residuals[: solver.log.num_iters + 1].copy()]
        res[key] = out
        ns.P_ns.close()
    for key in ((0, 1), (1, 0), (0, 0)):
        for a, c in zip(res[(1, 1)], res[key]):
            np.testing.assert_array_equal(a, c)
    np.testing.assert_array_equal(res[(1, 1)][0], orc.spmv(H["mats"][0], np.random.default_rng(7).uniform(-1, 1, n)))   # rows summed in the oracle's order
    for a, c in zip(res[(1, 1)][:3], res["dot"][:3]):
        np.testing.assert_array_equal(a, c)
    assert len(res["dot"][-1]) == len(res[(1, 1)][-1])
    np.testing.assert_allclose(res["dot"][-1], res[(1, 1)][-1], rtol=1e-9)
    assert rel_err(res["dot"][-2], res[(1, 1)][-2]) <= 1e-12
    fin = res[(1, 1)][1]
    assert not fin.all() and fin.sum() > 0.9 * n
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=[orc.Smoother(orc.JACOBI, 4, 2.0 / 3.0)] * (nlev - 1), maxiter=1)
    xo, nit, flag_o, hist = orc.cg_solve(H["mats"][0], b, x0=np.random.default_rng(8).uniform(-1, 1, n), Pl=go, maxiter=40, atol=1e-14, rtol=1e-8)
    assert len(res[(1, 1)][-1]) == nit + 1
    np.testing.assert_allclose(res[(1, 1)][-1], hist, rtol=1e-8)
    assert rel_err(res[(1, 1)][-2], xo) <= 1e-10


@pytest.mark.parametrize("nc,nlev", [((48, 48, 48), 3), ((132, 68), 3), ((34, 46, 30), 2)])
def test_pair_prolongation_is_bitwise_the_single_row_kernel(S, po, orc, nc, nlev):
    """sellp_pair_addto_kernel (dxh = P dxH ; xh += dxh with two rows per lane, levels of >= pat_r2mv_min rows) against sellp_kernel
    (option pat_pair_p = 0): V-cycles from random residuals (row counts that are not multiples of 128: the ragged last slice takes
    the element-wise path) agree to the last bit, and with the oracle's V-cycle to 1e-12."""
    H = po.build_hierarchy(nc, nlev, 1)
    n = H["mats"][0].shape[0]
    res = {}
    for pp in (1, 0):
        gmg = make_gmg(S, H, pre_smoothers=jac(S, nlev, 3), options={"pat_pair_p": pp, "pat_r2mv_min": 1})
        ns = setup(S, gmg, H["mats"][0])
        out = []
        for seed in (1, 2):
            z = np.zeros(n)
            S.solve_(z, ns, np.random.default_rng(seed).uniform(-1, 1, n))
            out.append(z)
        res[pp] = out
        ns.close()
    for a, c in zip(res[0], res[1]):
        np.testing.assert_array_equal(a, c)
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=[orc.Smoother(orc.JACOBI, 3, 2.0 / 3.0)] * (nlev - 1), maxiter=1)
    zo = go.solve(np.random.default_rng(1).uniform(-1, 1, n))[0]
    assert rel_err(res[1][0], zo) <= 1e-12


def test_work_issued_on_the_callers_stream_needs_no_synchronisation(S, po, orc):
    """gmg_set_stream: b is produced and x consumed by the caller's kernels on ITS stream, the V-cycle (gmg_apply on device vectors) is
    issued on the same stream, and nothing synchronises in between -- a long-running kernel queued in front of b's producer makes a
    missing ordering visible.  Same bits as the run on the handle's own stream with explicit synchronisation; back on the own stream
    (None) the handle keeps working."""
    import torch
    nc, nlev = (40, 40, 40), 3
    H = po.build_hierarchy(nc, nlev, 1)
    n = H["mats"][0].shape[0]
    gmg = make_gmg(S, H, pre_smoothers=jac(S, nlev, 4))
    ns = setup(S, gmg, H["mats"][0])
    base = torch.from_numpy(np.random.default_rng(5).uniform(-1, 1, n)).cuda()
    torch.cuda.synchronize()
    # reference: own stream, synchronised by hand
    b0 = (base * 3.0 + 1.0)
    x0 = torch.zeros(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    S.solve_(x0, ns, b0)
    torch.cuda.synchronize()
    ref = (x0 * 2.0).cpu().numpy()
    own = ns.get_stream()
    st = torch.cuda.Stream()
    ns.set_stream(st)
    assert ns.get_stream() == st.cuda_stream != own
    big = torch.empty(1 << 28, dtype=torch.float64, device="cuda")   # 2 GiB: the fill takes a few hundred microseconds
    with torch.cuda.stream(st):
        big.fill_(1.0)                                        # keeps the stream busy while the host runs ahead
        b1 = (base * 3.0 + 1.0)
        x1 = torch.zeros(n, dtype=torch.float64, device="cuda")
        S.solve_(x1, ns, b1)                                  # no synchronisation before ...
        out = x1 * 2.0                                        # ... or after
    st.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    ns.set_stream(None)
    assert ns.get_stream() == own
    x2 = torch.zeros(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    S.solve_(x2, ns, b0)
    torch.cuda.synchronize()
    np.testing.assert_array_equal((x2 * 2.0).cpu().numpy(), ref)
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=[orc.Smoother(orc.JACOBI, 4, 2.0 / 3.0)] * (nlev - 1), maxiter=1)
    zo = go.solve(b0.cpu().numpy())[0]
    assert rel_err(x2.cpu().numpy(), zo) <= 1e-12
    ns.close()


def test_streamed_matrix_of_an_own_ghost_level_needs_its_partition_first(S, po):
    """gmg_set_operator_rows on an own | ghost level splits every block at n_own: without the partition the shape n_own x (n_own +
    n_ghost) is rejected as a non-square level matrix, with it a block whose shape disagrees with the partition is rejected too."""
    import ctypes as C
    from gridapsolvers_jl_amd import abi
    lib = abi.load()
    h = C.c_void_p()
    abi.check(None, lib.gmg_create(C.byref(h), 2, 0))
    try:
        cb_x = abi.HOST_EXCHANGE_FN(lambda *a: None)
        cb_r = abi.HOST_ALLREDUCE_FN(lambda *a: None)
        abi.check(h, lib.gmg_comm_init_host(h, 0, 2, C.cast(cb_x, C.c_void_p), C.cast(cb_r, C.c_void_p), None))
        n_own, n_ghost = 6, 2
        ptr = np.arange(0, 2 * n_own + 1, 2, dtype=np.int64)
        idx = np.stack([np.arange(n_own), np.array([6, 7, 6, 7, 6, 7])], axis=1).reshape(-1).astype(np.int64)
        val = np.ones(idx.size)
        args = (C.c_void_p(ptr.ctypes.data), C.c_void_p(idx.ctypes.data), C.c_void_p(val.ctypes.data), 0, 8)
        rc = lib.gmg_set_operator_rows(h, 0, abi.OP_A, n_own, n_own + n_ghost, 0, n_own, *args)
        assert rc == abi.ERR_INVALID                          # no partition declared: not a square level matrix
        nbr = np.array([1], dtype=np.int32)
        sp, si, rp = np.array([0, 2], dtype=np.int64), np.array([0, 1], dtype=np.int64), np.array([0, 2], dtype=np.int64)
        abi.check(h, lib.gmg_set_partition(h, 0, n_own, n_ghost, 1, C.c_void_p(nbr.ctypes.data), C.c_void_p(sp.ctypes.data),
                                           C.c_void_p(si.ctypes.data), C.c_void_p(rp.ctypes.data)))
        rc = lib.gmg_set_operator_rows(h, 0, abi.OP_A, n_own, n_own + n_ghost + 1, 0, n_own, *args)
        assert rc == abi.ERR_INVALID                          # shape disagrees with the partition
        rc = lib.gmg_set_operator_rows(h, 0, abi.OP_A, n_own, n_own + n_ghost, 0, 3, *args)   # first half of the rows ...
        assert rc == abi.OK
        assert lib.gmg_set_operator_rows_repeat(h, 0, abi.OP_A, 3, 1, 3) == abi.ERR_UNSUPPORTED  # ... cannot be "repeated": ghost columns do not shift
        abi.check(h, lib.gmg_set_operator_rows(h, 0, abi.OP_A, n_own, n_own + n_ghost, 0, n_own, *args))
    finally:
        lib.gmg_destroy(h)
