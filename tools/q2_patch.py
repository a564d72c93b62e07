"""Config-3-shaped run (Q2, vertex-star patch smoother, FGMRES outer): python tools_q2.py [cells] [levels]"""
import sys, time, json, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.import_package(); po, S = pkg.poisson, pkg.solvers
nc = (int(sys.argv[1]),) * 3 if len(sys.argv) > 1 else (32,) * 3
nlev = int(sys.argv[2]) if len(sys.argv) > 2 else 3
order = 2
t0 = time.time(); H = po.build_hierarchy(nc, nlev, order); t_asm = time.time() - t0
t0 = time.time()
sm = []
for l in range(nlev - 1):
    pp, pd = po.vertex_star_patches(H["ncells"][l], order)
    sm.append(S.RichardsonSmoother(S.PatchSolver(pp, pd), 10, 0.2))
t_patch = time.time() - t0
b = po.dirichlet_lift_rhs(nc, order)
gmg = S.GMGLinearSolver(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=sm, post_smoothers=sm, maxiter=1)
solver = S.FGMRESSolver(5, gmg, maxiter=20, atol=1e-14, rtol=1e-6)
t0 = time.time(); ns = S.numerical_setup(S.symbolic_setup(solver, H["mats"][0]), H["mats"][0]); t_setup = time.time() - t0
bd = torch.from_numpy(b).cuda(); xd = torch.zeros_like(bd); torch.cuda.synchronize()
for _ in range(2):
    xd.zero_(); torch.cuda.synchronize(); S.solve_(xd, ns, bd)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(3):
    xd.zero_(); torch.cuda.synchronize(); S.solve_(xd, ns, bd)
torch.cuda.synchronize(); dt = (time.time() - t0) / 3
n = b.size
print(json.dumps(dict(cells=nc[0], levels=nlev, dofs=n, nnz=H["mats"][0].nnz, iters=solver.log.num_iters, flag=solver.log.flag,
                      hist=(solver.log.residuals[:solver.log.num_iters + 1] / solver.log.residuals[0]).tolist(),
                      ms_per_solve=dt * 1e3, dofs_per_s=n / dt, l2err=po.l2_error_sq(nc, order, xd.cpu().numpy()),
                      t_asm=t_asm, t_patch=t_patch, t_setup=t_setup, fmt=ns.P_ns.level_format(0), dev_GB=ns.P_ns.device_bytes() / 1e9)))
