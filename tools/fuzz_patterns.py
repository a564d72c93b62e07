"""Randomised check of the row-pattern layouts against the sequential SpMV / the oracle's literal Richardson loop.
python tools/fuzz_patterns.py [ncases] [seed]  (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import __graft_entry__ as entry
pkg = entry.import_package(); orc = entry.import_oracle()
po, S = pkg.poisson, pkg.solvers
from gridapsolvers_jl_amd import abi
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
seen = {}
for case in range(ncases):
    n = int(rng.choice([64, 65, 127, 200, 1000, 4097, 20000, 62 * 64, 62 * 64 + 1, 62 * 700 + 5]))
    noff = int(rng.integers(1, 40))
    stencil = bool(rng.integers(0, 3) == 0)                  # a third of the cases: runs of three consecutive offsets (shared-offset / one-launch paths)
    span = int(rng.choice([3, 10, n // 4 + 1, n // 2]))
    offs = np.unique(np.concatenate([[0], rng.integers(-span, span + 1, size=noff)]))
    if stencil:
        base = np.unique(rng.integers(-span, span + 1, size=int(rng.integers(1, 4))))
        offs = np.unique(np.concatenate([[-1, 0, 1]] + [[b - 1, b, b + 1] for b in base]))
    nval = int(rng.choice([1, 3, 1000]))                     # few distinct value sets -> few patterns ; many -> fallback layouts
    vals_sets = rng.uniform(-1, 1, size=(nval, offs.size)); vals_sets[:, offs == 0] = 5.0 + offs.size
    pick = rng.integers(0, nval, size=n)
    wrap = bool(rng.integers(0, 2))
    rows, cols, vals = [], [], []
    for k, o in enumerate(offs):
        c = np.arange(n) + o
        keep = np.ones(n, bool) if wrap else (c >= 0) & (c < n)
        rows.append(np.arange(n)[keep]); cols.append((c % n)[keep]); vals.append(vals_sets[pick, k][keep])
    A = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)); A.sum_duplicates(); A.sort_indices()
    nc = max(2, n // 3)
    Pm = sp.csr_matrix((np.ones(n), (np.arange(n), np.minimum(np.arange(n) // 3, nc - 1))), shape=(n, nc))
    Ac = sp.identity(nc, format="csr") * 2.0
    H = dict(mats=[po.CSR(A.shape, A.indptr, A.indices, A.data), po.CSR(Ac.shape, Ac.indptr, Ac.indices, Ac.data)],
             prolongations=[po.CSR(Pm.shape, Pm.indptr, Pm.indices, Pm.data)])
    niter, omega = int(rng.integers(1, 12)), float(rng.uniform(0.2, 0.9))
    sm = [S.RichardsonSmoother(S.JacobiLinearSolver(), niter, omega)]
    gmg = S.GMGLinearSolver(H["mats"], H["prolongations"], None, pre_smoothers=sm, post_smoothers=sm, maxiter=1)
    ns = S.numerical_setup(S.symbolic_setup(gmg, H["mats"][0]), H["mats"][0])
    x = rng.uniform(-1, 1, n); y = np.zeros(n)
    ns.op_apply(0, abi.OP_A, x, y)
    ok = np.array_equal(y, orc.spmv(H["mats"][0], x)) or np.max(np.abs(y - orc.spmv(H["mats"][0], x))) <= 1e-13 * np.max(np.abs(y))
    go = orc.GMG(H["mats"], H["prolongations"], pre_smoothers=[orc.Smoother(orc.JACOBI, niter, omega)], maxiter=1)
    x0, r0 = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    xs, rs = x0.copy(), r0.copy(); ns.smooth(0, xs, rs)
    xo, ro = go.smooth(0, x0, r0)
    ok2 = np.max(np.abs(xs - xo)) <= 1e-12 * max(1.0, np.max(np.abs(xo))) and np.max(np.abs(rs - ro)) <= 1e-12 * max(1.0, np.max(np.abs(ro)))
    fmt = ns.level_format(0)["layout"]
    seen[fmt] = seen.get(fmt, 0) + 1
    if not (ok and ok2):
        bad += 1
        print("MISMATCH", case, dict(n=n, noff=offs.size, span=span, nval=nval, wrap=wrap, niter=niter, fmt=fmt), flush=True)
    ns.close()
print(f"{ncases} cases, {bad} mismatches; layouts hit: {seen}")
sys.exit(1 if bad else 0)
