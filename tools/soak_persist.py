"""Soak test of the one-launch smoothing passes: N V-cycles (levels 1, 2 in one launch per pass) under background load on a second
stream, every result compared on the device with the per-sweep reference, bit for bit.  python tools/soak_persist.py [cells] [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as ge
pkg = ge.import_package()
S, po = pkg.solvers, pkg.poisson
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
nlev = 4
H = po.build_hierarchy((cells,) * 3, nlev, 1)
n = H["mats"][0].shape[0]
sm = [S.RichardsonSmoother(S.JacobiLinearSolver(), 10, 2.0 / 3.0)] * (nlev - 1)
mk = lambda: S.GMGLinearSolver(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=sm, post_smoothers=sm, maxiter=1)
rd = [torch.from_numpy(np.random.default_rng(k).uniform(-1, 1, n)).cuda() for k in range(8)]
os.environ["GMG_PERSIST"] = "0"
g0 = mk(); ns0 = S.numerical_setup(S.symbolic_setup(g0, H["mats"][0]), H["mats"][0])
ref = []
for r in rd:
    z = torch.zeros(n, dtype=torch.float64, device="cuda"); S.solve_(z, ns0, r); ref.append(z.clone())
os.environ["GMG_PERSIST"] = "1"
g1 = mk(); ns = S.numerical_setup(S.symbolic_setup(g1, H["mats"][0]), H["mats"][0])
side = torch.cuda.Stream()
a = torch.randn(4096, 4096, device="cuda"); b = torch.randn(4096, 4096, device="cuda"); small = torch.randn(128, 128, device="cuda")
z = torch.zeros(n, dtype=torch.float64, device="cuda")
bad = 0
torch.cuda.synchronize()
for it in range(N):
    if it % 3 != 2:
        with torch.cuda.stream(side):
            for k in range(1 + it % 4):
                (a @ b) if (it + k) % 3 else (small @ small)
    S.solve_(z, ns, rd[it % 8])
    if not torch.equal(z, ref[it % 8]):
        bad += 1
        print("MISMATCH at V-cycle", it, float((z - ref[it % 8]).abs().max()), flush=True)
torch.cuda.synchronize()
print(f"{N} V-cycles ({cells}^3, {nlev} levels; every smoothing pass of a level of <= 8192 slices is one launch): {bad} mismatches")
sys.exit(1 if bad else 0)
