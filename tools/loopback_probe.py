import sys, os, time, importlib
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import __graft_entry__ as e
pkg = e.import_package()
pa, mg, po, S = (importlib.import_module(pkg.__name__ + "." + m) for m in ("partition", "multigpu", "poisson", "solvers"))
cells, nlev = (96,) * 3, 4
H = po.build_hierarchy(cells, nlev, 1)
def unpart(tag):
    sm = [S.RichardsonSmoother(S.JacobiLinearSolver(), 10, 2.0 / 3.0)] * (nlev - 1)
    gm = S.GMGLinearSolver(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=sm, post_smoothers=sm, coarsest_solver=S.LUSolver(), maxiter=1,
                           mode="preconditioner", cycle_type="v_cycle")
    sol = S.CGSolver(gm, maxiter=20, atol=1e-14, rtol=1e-6)
    ns = S.numerical_setup(S.symbolic_setup(sol, H["mats"][0]), H["mats"][0])
    bd = torch.from_numpy(po.dirichlet_lift_rhs(cells, 1)).cuda(); xd = torch.zeros_like(bd)
    S.solve_(xd, ns, bd); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        xd.zero_(); torch.cuda.synchronize(); S.solve_(xd, ns, bd)
    torch.cuda.synchronize()
    print(tag, "ms per solve %.3f" % ((time.perf_counter() - t0) / 5 * 1e3), "iters", sol.log.num_iters, ns.P_ns.sweep_signature(0), flush=True)
    ns.P_ns.close()
unpart("before any communicator:")
grid = pa.rank_grid(8, 3)
F = pa.fold_ranks([pa.build_local_hierarchy(cells, nlev, grid, r, 1, None, 1, [0, 0, 0, 0], "jacobi") for r in range(8)])
for tr in ("host_loopback", "rccl_loopback"):
    g = mg.DistributedGMG(cells, nlev, 0, 2, device_id=0, transport=tr, local_hierarchy=F, cells_global=cells)
    b = torch.from_numpy(g.rhs_lin()).cuda(); x = torch.zeros(g.n_own, dtype=torch.float64, device="cuda")
    g.cg_solve(b, x, maxiter=20, atol=1e-14, rtol=1e-6); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        x.zero_(); torch.cuda.synchronize(); g.cg_solve(b, x, maxiter=20, atol=1e-14, rtol=1e-6)
    torch.cuda.synchronize()
    print(tr, "ms per solve %.3f" % ((time.perf_counter() - t0) / 5 * 1e3), g.sweep_signature(0), flush=True)
    g.close()
    unpart("after " + tr + ":")
