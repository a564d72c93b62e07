import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import __graft_entry__ as e
pkg = e.import_package(); orc = e.import_oracle()
S, po = pkg.solvers, pkg.poisson
import test_gpu_block as T
hier = lambda nc, nlev, order=1: po.build_hierarchy(nc, nlev, order)
X = T._stokes_like(S, po, orc, hier, (16, 16, 16), 3, True)
b = np.random.default_rng(5).uniform(-1, 1, X["n"])
solver = S.FGMRESSolver(20, X["Pd"], atol=1e-10, rtol=1e-12, maxiter=100)
ns = T.setup(S, solver, X["mat"])
x = np.zeros(X["n"]); S.solve_(x, ns, b)
xo, nit, flag, hist = orc.fgmres_solve(X["K"], b, Pr=X["Po"], m=20, maxiter=100, atol=1e-10, rtol=1e-12)
h = solver.log.residuals[:nit + 1]
print(nit, solver.log.num_iters)
for i in range(nit + 1):
    print(i, hist[i], h[i], abs(h[i] - hist[i]) / hist[i])
