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


def _solve(mg, F, cells, nlev, transport, W, env=None, options=None):
    import torch
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        g = mg.DistributedGMG(cells, nlev, 0, 2, device_id=0, transport=transport, local_hierarchy=F, cells_global=cells, options=options)
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
