"""Dense inverse of a LARGE coarsest level: python tools/coarse_timing.py [cells] [levels]  (Q1 Poisson; 64 cells / 2 levels puts
29 791 dofs -- the coarsest level of BASELINE config 3 -- on the coarsest level).  Runs the setup + a CG solve with the 64-wide
and with the 32-wide panels (GMG_GJ_WIDE_MIN) and compares the solutions; GMG_SETUP_TIMING=1 prints the library's own breakdown."""
import os, sys, time
os.environ.setdefault("GMG_SETUP_TIMING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.import_package()
S, po = pkg.solvers, pkg.poisson
import torch
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nlev = int(sys.argv[2]) if len(sys.argv) > 2 else 2
torch.zeros(1, device="cuda")
H = po.build_hierarchy((cells,) * 3, nlev, 1)
b = np.random.default_rng(0).standard_normal(H["mats"][0].shape[0])
sols = {}
for tag, wide_min in (("wide64", "1000"), ("panel32", "100000000")):
    os.environ["GMG_GJ_WIDE_MIN"] = wide_min
    sm = [S.RichardsonSmoother(S.JacobiLinearSolver(), 10, 2.0 / 3.0)] * (nlev - 1)
    gmg = S.GMGLinearSolver(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=sm, post_smoothers=sm, maxiter=1)
    solver = S.CGSolver(gmg, maxiter=30, atol=1e-14, rtol=1e-10)
    ss = S.symbolic_setup(solver, H["mats"][0])
    t0 = time.perf_counter()
    ns = S.numerical_setup(ss, H["mats"][0])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    x = np.zeros_like(b)
    S.solve_(x, ns, b)
    sols[tag] = x.copy()
    print(f"{tag}: coarse dofs {H['mats'][-1].shape[0]}, numerical_setup {1e3*(t1-t0):.1f} ms, iters {solver.log.num_iters}", flush=True)
    ns.P_ns.close()
d = np.linalg.norm(sols["wide64"] - sols["panel32"]) / np.linalg.norm(sols["panel32"])
print(f"relative difference of the two solutions: {d:.3e}")
assert d < 1e-9
