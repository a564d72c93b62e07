"""Where the first numerical setup of the bench problem goes: python tools/setup_timing.py [cells] (GMG_SETUP_TIMING=1 prints the
library's own breakdown on stderr)."""
import os, sys, time
os.environ.setdefault("GMG_SETUP_TIMING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.import_package()
S, po = pkg.solvers, pkg.poisson
import torch
_orig = S._set_op
def _timed(fn, h, lev, M):
    t0 = time.perf_counter()
    _orig(fn, h, lev, M)
    print(f"    {fn.__name__:22s} level {lev}: {1e3 * (time.perf_counter() - t0):7.1f} ms  ({M.shape[0]} rows, {M.nnz} nnz)", flush=True)
S._set_op = _timed
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nlev = 4 if cells <= 128 else 5
torch.zeros(1, device="cuda")
H = po.build_hierarchy((cells,) * 3, nlev, 1, kappa=po.smooth_kappa if "varcoef" in sys.argv else None)
for rep in range(3):
    sm = [S.RichardsonSmoother(S.JacobiLinearSolver(), 10, 2.0 / 3.0)] * (nlev - 1)
    t0 = time.perf_counter()
    gmg = S.GMGLinearSolver(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=sm, post_smoothers=sm, maxiter=1)
    solver = S.CGSolver(gmg, maxiter=30, atol=1e-14, rtol=1e-6)
    t1 = time.perf_counter()
    ss = S.symbolic_setup(solver, H["mats"][0])
    t2 = time.perf_counter()
    ns = S.numerical_setup(ss, H["mats"][0])
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"rep {rep}: constructors {1e3*(t1-t0):.1f} ms, symbolic_setup {1e3*(t2-t1):.1f} ms, numerical_setup {1e3*(t3-t2):.1f} ms", flush=True)
    ns.P_ns.close()
