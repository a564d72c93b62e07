"""numerical_setup! with new values on the same sparsity (value refresh) against a fresh setup, variable-coefficient 3-D Poisson:
python tools/refresh_timing.py [cells]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.import_package()
S, po = pkg.solvers, pkg.poisson
import torch
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nlev = 4 if cells <= 128 else 5
torch.zeros(1, device="cuda")
k2 = lambda X, Y, Z: 2.0 + np.cos(3.0 * X) * np.sin(2.0 * Y + 0.2) + 0.5 * Z * Z
H1 = po.build_hierarchy((cells,) * 3, nlev, 1, kappa=po.smooth_kappa)
H2 = po.build_hierarchy((cells,) * 3, nlev, 1, kappa=k2)
sm = [S.RichardsonSmoother(S.JacobiLinearSolver(), 10, 2.0 / 3.0)] * (nlev - 1)
mk = lambda H: S.CGSolver(S.GMGLinearSolver(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=sm, post_smoothers=sm, maxiter=1),
                          maxiter=30, atol=1e-14, rtol=1e-6)
solver = mk(H1)
t0 = time.perf_counter(); ns = S.numerical_setup(S.symbolic_setup(solver, H1["mats"][0]), H1["mats"][0]); torch.cuda.synchronize()
print(f"first setup            {1e3 * (time.perf_counter() - t0):8.1f} ms")
uex = po.nodal_values((cells,) * 3, 1)
for rep in range(3):
    Hn = H2 if rep % 2 == 0 else H1
    t0 = time.perf_counter(); S.numerical_setup_(ns, Hn["mats"][0], Hn["mats"]); torch.cuda.synchronize()
    t_ref = time.perf_counter() - t0
    b = Hn["mats"][0].matvec(uex); x = np.zeros_like(b)
    S.solve_(x, ns, b)
    print(f"refresh (all levels)   {1e3 * t_ref:8.1f} ms   -> {solver.log.num_iters} CG iterations, max error {np.max(np.abs(x - uex)):.2e}")
s2 = mk(H2)
t0 = time.perf_counter(); ns2 = S.numerical_setup(S.symbolic_setup(s2, H2["mats"][0]), H2["mats"][0]); torch.cuda.synchronize()
print(f"fresh setup, new values {1e3 * (time.perf_counter() - t0):7.1f} ms")
