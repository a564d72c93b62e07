#!/usr/bin/env python3
"""BASELINE config 3: 3-D Poisson Q2, 5-level GMG with the vertex-star patch smoother Richardson(PatchSolver,10,0.2)
pre = post on every level, FGMRES(5) outer, rtol 1e-6 (test/LinearSolvers/GMGTests.jl:18-47,119-123).

    python tools/config3.py [--cells 256] [--levels 5] [--steps 2] [--warmup 1] [--stream-min-rows 1000000]

Operators with at least --stream-min-rows rows are generated and handed to the library as row-block streams
(gmg_set_operator_rows): neither the driver nor the library ever holds their CSR (8.5e9 nonzeros at 256^3).
Prints one JSON object: iteration count, residual history, time per solve, DoFs/s, L2 error (the reference tests'
criterion), true residual through the device operator, setup breakdown, device memory."""
import argparse
import json
import os
import resource
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, default=256)
ap.add_argument("--levels", type=int, default=5)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--warmup", type=int, default=1)
ap.add_argument("--stream-min-rows", type=int, default=1000000)
ap.add_argument("--no-l2", action="store_true")
ap.add_argument("--profile", action="store_true", help="cProfile of numerical_setup (top 30 by cumulative time, stderr)")
a = ap.parse_args()

pkg = entry.import_package()
po, S, abi = pkg.poisson, pkg.solvers, pkg.abi
nc, nlev, order = (a.cells,) * 3, a.levels, 2
t0 = time.time()
H = po.build_hierarchy(nc, nlev, order, stream_min_rows=a.stream_min_rows)
t_asm = time.time() - t0                       # small levels only: streamed operators are generated inside numerical_setup
t0 = time.time()
sm = []
npatch = []
for l in range(nlev - 1):
    pp, pd = po.vertex_star_patches(H["ncells"][l], order)
    npatch.append(int(pp.size - 1))
    sm.append(S.RichardsonSmoother(S.PatchSolver(pp, pd), 10, 0.2))
t_patch = time.time() - t0
t0 = time.time()
b = po.dirichlet_lift_rhs(nc, order)
t_rhs = time.time() - t0
gmg = S.GMGLinearSolver(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=sm, post_smoothers=sm, maxiter=1)
solver = S.FGMRESSolver(5, gmg, maxiter=20, atol=1e-14, rtol=1e-6)
os.environ.setdefault("GMG_SETUP_TIMING", "1")
t0 = time.time()
if a.profile:
    import cProfile
    import pstats
    pr = cProfile.Profile()
    ns = pr.runcall(S.numerical_setup, S.symbolic_setup(solver, H["mats"][0]), H["mats"][0])
    pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(30)
else:
    ns = S.numerical_setup(S.symbolic_setup(solver, H["mats"][0]), H["mats"][0])
t_setup = time.time() - t0
del sm, gmg.pre_smoothers[:], gmg.post_smoothers[:]
bd = torch.from_numpy(b).cuda()
xd = torch.zeros_like(bd)
torch.cuda.synchronize()
for _ in range(a.warmup):
    xd.zero_(); torch.cuda.synchronize(); S.solve_(xd, ns, bd)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(a.steps):
    xd.zero_(); torch.cuda.synchronize(); S.solve_(xd, ns, bd)
torch.cuda.synchronize()
dt = (time.time() - t0) / a.steps
n = b.size
# true residual through the device operator (mul!(y, A, x) of the streamed level matrix)
yd = torch.zeros_like(bd)
ns.P_ns.op_apply(0, abi.OP_A, xd, yd)
true_rel = float(torch.linalg.vector_norm(bd - yd) / torch.linalg.vector_norm(bd))
x = xd.cpu().numpy()
out = dict(workload=f"BASELINE configs[2]: 3D Poisson Q2 {a.cells}^3 cells, {nlev}-level GMG, Richardson(PatchSolver,10,0.2) pre=post, "
                    f"FGMRES(5) rtol=1e-6, rhs = u=x1+x2 Dirichlet lift",
           cells=a.cells, levels=nlev, dofs=int(n), dofs_per_level=[int(M.shape[0]) for M in H["mats"]], patches_per_level=npatch,
           streamed_levels=[l for l, M in enumerate(H["mats"]) if hasattr(M, "row_blocks")],
           iters=int(solver.log.num_iters), flag=int(solver.log.flag),
           hist_rel=(solver.log.residuals[:solver.log.num_iters + 1] / solver.log.residuals[0]).tolist(),
           ms_per_solve=dt * 1e3, dofs_per_s=n / dt, true_residual_rel=true_rel,
           l2_error_sq=None if a.no_l2 else po.l2_error_sq(nc, order, x), max_abs_error=float(np.max(np.abs(x - po.nodal_values(nc, order)))),
           t_small_levels_asm=t_asm, t_patch_tables=t_patch, t_rhs=t_rhs, t_numerical_setup_incl_stream_generation=t_setup,
           finest_format=ns.P_ns.level_format(0), device_GB=ns.P_ns.device_bytes() / 1e9,
           host_peak_rss_GB=resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6)
print(json.dumps(out))
