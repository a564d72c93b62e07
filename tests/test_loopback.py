"""The product's exchange path over RCCL on ONE GPU: a W-rank partition folded onto one rank whose every
neighbour is itself (partition.fold_ranks + gmg_comm_set_loopback).  pack -> ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on the
communication stream -> event -> boundary fix-up -> ncclAllReduce run through gmg_cg_solve exactly as on W GPUs.  Stands for the
reference's consistent!(::PVector) around the local work (PatchSolvers.jl:227-236, JacobiLinearSolvers.jl:49-56) and its dot / norm
over parts (CGSolvers.jl:85-111); the reference's own bar for the distributed path is its MPI test run with 1 and 4 processes
(test/LinearSolvers/mpi/GMGTests.jl:5-8)."""
import importlib
import os

import numpy as np
import pytest

from conftest import rel_err


def _mods(pkg):
    return importlib.import_module(pkg.__name__ + ".partition"), importlib.import_module(pkg.__name__ + ".multigpu")


def _folded(pa, cells, nlev, W, depth=None, rep_from=None, order=1):
    grid = pa.rank_grid(W, len(cells))
    return pa.fold_ranks([pa.build_local_hierarchy(cells, nlev, grid, r, order, None, rep_from, depth, "jacobi") for r in range(W)])


def _fill(L, x_own):
    """the folded local vector of a level from its owned entries: consistent!(v) through the level's self-exchange plan"""
    if getattr(L, "overlap", False):
        v = np.zeros(L.n_local)
        v[L.own_idx] = x_own
        v[L.rcv_idx] = v[L.snd_idx]
        return v
    return np.concatenate([x_own, x_own[L.snd_idx]])


def _own(L, v):
    return v[L.own_idx] if getattr(L, "overlap", False) else v[: L.n_own]


@pytest.mark.parametrize("cells,nlev,W,depth,rep", [((16, 16, 16), 3, 8, None, None), ((32, 32), 4, 4, [0, 3, 5, 0], 3),
                                                     ((24, 16, 16), 3, 2, [0, 2, 0], 2), ((16, 16, 16), 4, 8, None, 2)])
def test_folded_partition_is_the_global_hierarchy(pkg, po, cells, nlev, W, depth, rep):
    """owned rows of every folded operator applied to a consistent folded vector = the global operator on the global vector"""
    pa, _ = _mods(pkg)
    F = _folded(pa, cells, nlev, W, depth, rep)
    H = po.build_hierarchy(cells, nlev, 1)
    rng = np.random.default_rng(7)
    lv = F["levels"]
    for l, L in enumerate(lv):
        n = H["mats"][l].shape[0]
        assert np.array_equal(np.sort(L.own_gid), np.arange(n))                   # every dof owned exactly once
        xg = rng.standard_normal(n)
        if L.replicated:
            assert L.A.shape == (n, n)
            continue
        assert np.all(L.nbr_rank == 1) and L.snd_ptr[-1] == L.rcv_ptr[-1] == L.n_ghost
        v = _fill(L, xg[L.own_gid])
        y = L.A.matvec(v)
        assert np.max(np.abs(_own(L, y) - H["mats"][l].matvec(xg)[L.own_gid])) < 1e-12 * np.max(np.abs(y))
        if l + 1 < nlev:
            C = lv[l + 1]
            nc_ = H["mats"][l + 1].shape[0]
            xc = rng.standard_normal(nc_)
            vc = xc if C.replicated else _fill(C, xc[C.own_gid])
            p = L.P.matvec(vc)
            assert np.max(np.abs(_own(L, p) - H["prolongations"][l].matvec(xc)[L.own_gid])) < 1e-13
            r = L.R.matvec(v)
            want = H["restrictions"][l].matvec(xg)
            if C.replicated:
                assert np.max(np.abs(r - want[F["rep_gid"]])) < 1e-12 and np.array_equal(np.sort(F["rep_gid"]), np.arange(nc_))
            else:
                assert np.max(np.abs(_own(C, r) - want[C.own_gid])) < 1e-12


def _solve(mg, F, cells, nlev, transport, W, env=None, options=None, cycle_type="v_cycle"):
    import torch
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        g = mg.DistributedGMG(cells, nlev, 0, 2, device_id=0, transport=transport, local_hierarchy=F, cells_global=cells, options=options,
                              cycle_type=cycle_type)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    b = torch.from_numpy(g.rhs_lin()).cuda()
    x = torch.zeros(g.n_own, dtype=torch.float64, device="cuda")
    ex0, ar0 = g.comm_stats()
    log = g.cg_solve(b, x, maxiter=20, atol=1e-14, rtol=1e-6)
    torch.cuda.synchronize()
    ex1, ar1 = g.comm_stats()
    info = g.comm_info()
    out = dict(x=x.cpu().numpy(), iters=log.num_iters, hist=np.array(log.residuals[: log.num_iters + 1]), exchanges=ex1 - ex0,
               allreduces=ar1 - ar0, info=info, exact=g.exact_own(), gid=F["levels"][0].own_gid, sig=g.sweep_signature(0))
    g.close()
    return out


@pytest.mark.gpu
@pytest.mark.child_process
@pytest.mark.parametrize("cells,nlev,W,depth,rep", [((32, 32, 32), 4, 8, None, None), ((32, 32, 32), 4, 8, [0, 3, 0, 0], 2),
                                                     ((48, 32, 32), 3, 2, None, None)])
def test_cg_gmg_over_rccl_loopback_is_bitwise_the_host_transport_and_matches_the_oracle(pkg, po, orc, cells, nlev, W, depth, rep):
    """The partitioned CG + GMG solve with every halo exchange and every all-reduce going through RCCL (one real rank, self messages),
    overlapped with the own x own kernels and in-stream, against (a) the same folded partition through the host-staged loopback: the
    same bits, the same exchange count; (b) the serial oracle on the global hierarchy: same iteration count, history to 1e-10,
    solution to 1e-10 (the own | ghost split changes the order of the row sums, nothing else)."""
    pa, mg = _mods(pkg)
    F = _folded(pa, cells, nlev, W, depth, rep)
    H = po.build_hierarchy(cells, nlev, 1)
    b = po.dirichlet_lift_rhs(cells, 1)
    xo, nit, flag, hist = orc.cg_solve(H["mats"][0], b, Pl=orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=1),
                                       maxiter=20, atol=1e-14, rtol=1e-6)
    host = _solve(mg, F, cells, nlev, "host_loopback", W)
    runs = {"overlapped": _solve(mg, F, cells, nlev, "rccl_loopback", W, env={"GMG_OVERLAP": "1"}),
            "in_stream": _solve(mg, F, cells, nlev, "rccl_loopback", W, env={"GMG_OVERLAP": "0"})}
    assert host["info"]["transport"] == "host" and host["info"]["nranks"] == 2
    for name, r in runs.items():
        assert r["info"]["transport"] == "rccl" and r["info"]["rccl_comm_count"] == 1 and r["info"]["nranks"] == 2, (name, r["info"])
        assert r["exchanges"] == host["exchanges"] > 20 * r["iters"] and r["allreduces"] == host["allreduces"] >= 2 * r["iters"], (name, r)
        assert r["iters"] == host["iters"] == nit, (name, r["iters"], host["iters"], nit)
        assert np.array_equal(r["x"], host["x"]) and np.array_equal(r["hist"], host["hist"]), name
        assert np.max(np.abs(r["hist"] - hist) / hist) < 1e-10 and rel_err(r["x"], xo[r["gid"]]) < 1e-10, name
        assert np.max(np.abs(r["x"] - r["exact"])) < 1e-5


@pytest.mark.gpu
@pytest.mark.child_process
def test_generic_layout_over_rccl_loopback(pkg, po, orc):
    """the same with every structure-exploiting layout off (SELL-64 own x own stream, s-carrying halo, fused pack in the boundary fix-up)"""
    pa, mg = _mods(pkg)
    cells, nlev, W = (32, 32, 32), 3, 8
    F = _folded(pa, cells, nlev, W)
    H = po.build_hierarchy(cells, nlev, 1)
    b = po.dirichlet_lift_rhs(cells, 1)
    xo, nit, flag, hist = orc.cg_solve(H["mats"][0], b, Pl=orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=1),
                                       maxiter=20, atol=1e-14, rtol=1e-6)
    opts = {"vdict": 0, "idx16": 0, "pattern": 0, "opattern": 0}
    host = _solve(mg, F, cells, nlev, "host_loopback", W, options=opts)
    r = _solve(mg, F, cells, nlev, "rccl_loopback", W, options=opts)
    assert r["iters"] == host["iters"] == nit and np.array_equal(r["x"], host["x"]) and r["exchanges"] == host["exchanges"]
    assert rel_err(r["x"], xo[r["gid"]]) < 1e-10 and "sell" in r["sig"]


@pytest.mark.gpu
def test_bench_prints_exactly_one_line_on_stdout_even_when_rccl_prints_its_banner(tmp_path):
    """NCCL_DEBUG=VERSION (set on the GPU boxes) makes RCCL print a five-line version banner to stdout through C stdio -- at process
    exit when stdout is a pipe, i.e. BEHIND the JSON line of any bench run that creates a communicator.  bench.py points file
    descriptor 1 at stderr and writes its one line to a duplicate of the original stdout (claim_stdout): the driver's parser sees one line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NCCL_DEBUG="VERSION", GMG_BENCH_DETAILS=str(tmp_path / "legs.json"), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--legs", "default,rccl_loopback", "--cells", "32", "--levels", "3",
                          "--loopback-cells", "16", "--loopback-levels", "3", "--steps", "2", "--warmup", "1"],
                         env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    lines = out.stdout.splitlines()
    assert len(lines) == 1 and len(lines[0]) <= 4096, out.stdout[-800:]
    d = json.loads(lines[0])
    lb = d["rccl_loopback"]
    assert lb["rccl_ranks"] == 1 and lb["virtual_ranks"] == 8 and lb["bitwise_equal_host_transport"] is True and lb["exchanges_per_solve"] > 20
    assert "RCCL version" in out.stderr          # (the banner exists -- and went to stderr)


@pytest.mark.gpu
@pytest.mark.child_process
@pytest.mark.parametrize("cells,W", [((8, 8, 8), 8), ((12, 8, 8), 2)])
def test_patch_smoother_consistent_and_assemble_over_rccl_loopback(pkg, po, orc, cells, W):
    """The distributed patch smoother (PatchSolvers.jl:227-258: consistent!(r) -> local patch solves -> assemble!(dx) -> consistent!(dx) inside
    r -= A dx) with BOTH directions of the halo over RCCL: Q2, vertex-star patches owned per rank with driver-assembled matrices, FGMRES(5)
    outside, 2 x 2 x 2 and 2 x 1 x 1 partitions folded onto the GPU.  Bit-identical to the host-staged loopback; iteration count and solution
    of the serial oracle."""
    pa, mg = _mods(pkg)
    import torch
    nlev, order = 2, 2
    F = _folded(pa, cells, nlev, W, order=order)
    tabs = pa.fold_patch_tables(F)
    H = po.build_hierarchy(cells, nlev, order)
    b = po.dirichlet_lift_rhs(cells, order)
    osm = [orc.Smoother(orc.PATCH, 5, 0.2, *po.vertex_star_patches(H["ncells"][l], order)) for l in range(nlev - 1)]
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=osm, maxiter=1)
    xo, nit, flag, hist = orc.fgmres_solve(H["mats"][0], b, Pr=go, m=5, maxiter=30, atol=1e-14, rtol=1e-8)
    res = {}
    for tr in ("host_loopback", "rccl_loopback"):
        g = mg.DistributedGMG(cells, nlev, 0, 2, device_id=0, transport=tr, local_hierarchy=F, cells_global=cells, order=order, smoother="patch",
                              niter=5, omega=0.2, patch_tables=tabs)
        bb = torch.from_numpy(g.rhs_lin()).cuda()
        x = torch.zeros(g.n_own, dtype=torch.float64, device="cuda")
        ex0 = g.comm_stats()[0]
        log = g.fgmres_solve(bb, x, m=5, maxiter=30, atol=1e-14, rtol=1e-8)
        torch.cuda.synchronize()
        res[tr] = (x.cpu().numpy(), log.num_iters, np.array(log.residuals[: log.num_iters + 1]), g.comm_stats()[0] - ex0, g.comm_info())
        g.close()
    h, r = res["host_loopback"], res["rccl_loopback"]
    assert r[4]["transport"] == "rccl" and r[4]["rccl_comm_count"] == 1
    assert r[1] == h[1] == nit and r[3] == h[3] > 10 * nit
    assert np.array_equal(r[0], h[0]) and np.array_equal(r[2], h[2])
    gid = F["levels"][0].own_gid
    assert np.max(np.abs(r[2] - hist) / hist) < 1e-8 and rel_err(r[0], xo[gid]) < 1e-9


@pytest.mark.gpu
@pytest.mark.child_process
@pytest.mark.parametrize("n,nlev,W", [(16, 3, 4), (8, 2, 2)])
def test_distributed_stokes_block_solver_over_rccl_loopback(pkg, po, orc, n, nlev, W):
    """BASELINE configs[4] in its multi-rank form (test/Applications/mpi/StokesGMG.jl:5-12) with every exchange over RCCL, on one GPU: the
    lid-driven-cavity Q2 / P1disc system partitioned by cell boxes over W ranks (dpartition.py) and folded onto one (FoldedSpace) --
    vector-valued velocity GMG(maxiter 4) with owned vertex-star patch smoothers (driver-assembled matrices, consistent! + assemble!), the
    DISTRIBUTED patch-corrected prolongation with the grad-div rhs form, CG-Jacobi pressure block, upper block-triangular preconditioner
    (BlockTriangularSolvers.jl:216-242 on block vectors), FGMRES(20); the block handle's own communicator is a loopback one too
    (gmg_block_comm_set_loopback).  Bit-identical to the host-staged loopback; iteration count, history and solution of the serial oracle."""
    import importlib
    import torch
    pa, mg = _mods(pkg)
    st, dp = (importlib.import_module(pkg.__name__ + "." + m) for m in ("stokes", "dpartition"))
    alpha = 1.0e3
    grid = pa.rank_grid(W, 2)
    sysd, Hv = st.stokes_system(n, alpha), st.velocity_hierarchy(n, nlev, alpha)
    cells, npart = Hv["ncells"], nlev - 1
    V = [dp.Space(f"v{l}", st.velocity_owner(cells[l], grid), W) for l in range(nlev)]
    Pq = dp.Space("p", st.pressure_owner(n, grid), W)
    A = sysd["A"]
    sc = lambda M: M.to_scipy()
    ops = [(sc(A[0][0]), V[0], V[0]), (sc(A[0][1]), V[0], Pq), (sc(A[1][0]), Pq, V[0]), (sc(sysd["Mp_scaled"]), Pq, Pq)]
    pats, star_own, int_own = [], [], []
    for l in range(npart):
        ops += [(sc(Hv["mats"][l]), V[l], V[l]), (sc(Hv["graddiv"][l]), V[l], V[l]), (sc(Hv["restrictions"][l]), V[l + 1], V[l])]
        if l + 1 < npart:
            ops.append((sc(Hv["prolongations"][l]), V[l], V[l + 1]))
        star_own.append(st.patch_owner(*Hv["star_patches"][l], V[l].owner))
        int_own.append(st.patch_owner(*Hv["interior_patches"][l], V[l].owner))
        pats += [(*Hv["star_patches"][l], V[l], star_own[l]), (*Hv["interior_patches"][l], V[l], int_own[l])]
    dp.partition_spaces(V + [Pq], ops, pats)
    FV, FP = [dp.FoldedSpace(V[l]) for l in range(npart)], dp.FoldedSpace(Pq)
    R_ = range(W)

    class LL:
        overlap = False

    def fold_patches(tabs, space, F, owners, Ag=None):
        ptr, loc, blocks = [np.zeros(1, dtype=np.int64)], [], []
        for r in R_:
            p_, l_, _g, b_ = dp.local_patches(*tabs, space, owners, r, Ag)
            ptr.append(p_[1:] + ptr[-1][-1]); loc.append(F.cmap[r][l_])
            if b_ is not None:
                blocks.append(b_)
        return np.concatenate(ptr), np.concatenate(loc).astype(np.int64), (np.ascontiguousarray(np.concatenate(blocks)) if blocks else None)
    levels, ptabs, ctabs = [], [], []
    rep_gid = np.zeros(0, dtype=np.int64)
    for l in range(nlev):
        L = LL()
        if l < npart:
            F = FV[l]
            F.plan_into(L)
            nc = F.n_own + F.n_ghost
            L.replicated = False
            L.own_gid = F.own_gid
            L.A = dp.stack_rows([dp.local_operator(sc(Hv["mats"][l]), V[l], V[l], r) for r in R_], F.cmap, nc)
            if l + 1 < npart:
                Fc = FV[l + 1]
                L.P = dp.stack_rows([dp.local_operator(sc(Hv["prolongations"][l]), V[l], V[l + 1], r) for r in R_], Fc.cmap, Fc.n_own + Fc.n_ghost)
                L.R = dp.stack_rows([dp.local_operator(sc(Hv["restrictions"][l]), V[l + 1], V[l], r) for r in R_], F.cmap, nc)
            else:                                               # boundary to the replicated level: global coarse columns, the ranks' coarse rows stacked
                Pg = sc(Hv["prolongations"][l]).tocsr()[F.own_gid]
                Pg.sort_indices()
                L.P = po.CSR(Pg.shape, Pg.indptr.astype(np.int64), Pg.indices.astype(np.int32), Pg.data)
                L.R = dp.stack_rows([dp.local_operator(sc(Hv["restrictions"][l]), V[l + 1], V[l], r) for r in R_], F.cmap, nc)
                rep_gid = np.concatenate([V[l + 1].own[r] for r in R_])
            ptabs.append(fold_patches(Hv["star_patches"][l], V[l], F, star_own[l], sc(Hv["mats"][l])))
            cptr, cloc, _ = fold_patches(Hv["interior_patches"][l], V[l], F, int_own[l])
            ctabs.append((cptr, cloc, dp.stack_rows([dp.local_operator(sc(Hv["graddiv"][l]), V[l], V[l], r) for r in R_], F.cmap, nc)))
        else:
            L.replicated = True
            L.A = Hv["mats"][l]
            L.n_own, L.n_ghost = L.A.shape[0], 0
            ptabs.append(None); ctabs.append(None)
        levels.append(L)
    local = dict(levels=levels, rep_from=npart, rep_gid=np.ascontiguousarray(rep_gid, dtype=np.int64), cells=[(c, c) for c in cells],
                 grid=grid, order=2, rank=0, nranks=2)
    lev1 = FP.plan_into(LL())
    A01 = dp.stack_rows([dp.local_operator(sc(A[0][1]), V[0], Pq, r) for r in R_], FP.cmap, FP.n_own + FP.n_ghost)
    A10 = dp.stack_rows([dp.local_operator(sc(A[1][0]), Pq, V[0], r) for r in R_], FV[0].cmap, FV[0].n_own + FV[0].n_ghost)
    M11 = dp.stack_rows([dp.local_operator(sc(sysd["Mp_scaled"]), Pq, Pq, r) for r in R_], FP.cmap, FP.n_own + FP.n_ghost)
    nu, npp = sysd["sizes"]
    bg = sysd["b"]
    b = np.concatenate([bg[:nu][FV[0].own_gid], bg[nu:][FP.own_gid]])
    rg = np.random.default_rng(2).uniform(-1, 1, nu)
    res = {}
    for tr in ("host_loopback", "rccl_loopback"):
        g = mg.DistributedGMG((n // grid[0], n // grid[1]), nlev, 0, 2, device_id=0, transport=tr, order=2, niter=10, omega=0.2, gmg_maxiter=4,
                              gmg_rtol=1e-8, local_hierarchy=local, smoother="patch", patch_tables=ptabs, pcorr_tables=ctabs, cells_global=(n, n))
        z = np.zeros(FV[0].n_own)
        glog = g.apply(np.ascontiguousarray(rg[FV[0].own_gid]), z, maxiter=4)
        blk = mg.DistributedBlockSolver(g, A01, A10, M11, levels[0], lev1, A11=None, coeffs=((1.0, 1.0), (0.0, 1.0)), half="upper", cg=(20, 1e-14, 1e-6))
        x = np.zeros_like(b)
        ex0 = g.comm_stats()[0]
        log = blk.fgmres_solve(b, x, m=20, maxiter=100, atol=1e-10, rtol=1e-12)
        res[tr] = (x, z, int(log.num_iters), np.array(log.residuals[: log.num_iters + 1]), int(glog.num_iters), g.comm_stats()[0] - ex0, g.comm_info())
        blk.close(); g.close()
    h, r = res["host_loopback"], res["rccl_loopback"]
    assert r[6]["transport"] == "rccl" and r[6]["rccl_comm_count"] == 1
    assert np.array_equal(r[0], h[0]) and np.array_equal(r[1], h[1]) and np.array_equal(r[3], h[3]) and r[2] == h[2] and r[5] == h[5] > 100
    osm = [orc.Smoother(orc.PATCH, 10, 0.2, pp, pd) for pp, pd in Hv["star_patches"]]

    def make_go():
        return orc.GMG(Hv["mats"], Hv["prolongations"], Hv["restrictions"], pre_smoothers=osm, maxiter=4, rtol=1e-8,
                       prolongation_patches=[(orc.PATCH, *Hv["interior_patches"][l], Hv["graddiv"][l]) for l in range(nlev - 1)])
    zo, nit_g, _, _ = make_go().solve(rg)
    Po = orc.BlockPreconditioner([nu, npp], [make_go(), (orc.BD_CG_JACOBI, sysd["Mp_scaled"], 20, 1e-14, 1e-6)],
                                 {(0, 1): (A[0][1], 1.0), (1, 0): (A[1][0], 0.0)}, orc.UPPER)
    K = sysd["K"]
    xo, nit, flag, hist = orc.fgmres_solve(po.CSR(K.shape, K.indptr, K.indices, K.data), bg, Pr=Po, m=20, maxiter=100, atol=1e-10, rtol=1e-12)
    xg, zg = np.zeros(nu + npp), np.zeros(nu)
    xg[FV[0].own_gid] = r[0][: nu]; xg[nu + FP.own_gid] = r[0][nu:]
    zg[FV[0].own_gid] = r[1]
    assert r[4] == nit_g and rel_err(zg, zo) < 1e-8
    assert r[2] == nit and rel_err(xg, xo) < 1e-6 and np.max(np.abs(r[3] - hist) / hist[0]) < 1e-6
    assert np.linalg.norm(K @ xg - bg) < 1e-7


@pytest.mark.gpu
@pytest.mark.child_process
@pytest.mark.parametrize("cycle", ["w_cycle", "f_cycle"])
def test_w_and_f_cycles_over_rccl_loopback(pkg, po, orc, cycle):
    """gmg_w_cycle! / gmg_f_cycle! (GMGLinearSolvers.jl:504-610) on a partitioned hierarchy -- two partitioned levels, the second leg of every
    level re-smoothing and restricting again -- with every exchange and all-reduce over RCCL: the same bits as the host-staged loopback, the
    oracle's iteration count and solution."""
    pa, mg = _mods(pkg)
    cells, nlev, W = (32, 32, 32), 4, 8
    F = _folded(pa, cells, nlev, W, None, 2)                      # levels 0 and 1 partitioned, 2 and 3 replicated
    H = po.build_hierarchy(cells, nlev, 1)
    b = po.dirichlet_lift_rhs(cells, 1)
    go = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=1, cycle={"w_cycle": orc.W_CYCLE, "f_cycle": orc.F_CYCLE}[cycle])
    xo, nit, flag, hist = orc.cg_solve(H["mats"][0], b, Pl=go, maxiter=20, atol=1e-14, rtol=1e-6)
    host = _solve(mg, F, cells, nlev, "host_loopback", W, cycle_type=cycle)
    r = _solve(mg, F, cells, nlev, "rccl_loopback", W, cycle_type=cycle)
    assert r["iters"] == host["iters"] == nit and r["exchanges"] == host["exchanges"] and r["allreduces"] == host["allreduces"]
    assert np.array_equal(r["x"], host["x"]) and np.array_equal(r["hist"], host["hist"])
    assert np.max(np.abs(r["hist"] - hist) / hist) < 1e-10 and rel_err(r["x"], xo[r["gid"]]) < 1e-10
