#!/usr/bin/env python3
"""cpu_baseline leg of bench.py: times the CPU oracle (oracle/gmg_oracle.c, a cited restatement of the reference
algorithm -- kind "port") on the bench workload, in its OWN process.

TEST / MEASUREMENT INFRASTRUCTURE, not product code.  bench.py starts this script as a child process (never imports it)
because (1) the OpenMP runtime reads OMP_NUM_THREADS / OMP_PROC_BIND when it is loaded -- inside bench.py PyTorch has
already loaded and configured one -- and (2) the parent holds the GPU.  The child never touches the GPU.

Two timings on a bounded sample of full CG+GMG solves of the same workload:
  * one thread -- the analogue of ONE reference MPI rank (the reference is single-threaded per rank);
  * all usable host cores (OpenMP build of the same file) -- the analogue of the reference under MPI on P ranks.
    usable cores = min(len(os.sched_getaffinity(0)), cgroup cpu quota); threads are pinned (OMP_PROC_BIND=close,
    OMP_PLACES=cores) and the operator arrays are first-touched in parallel (orc_parallel_copy).
Writes one JSON object to stdout and the single-thread solution to --out-x (npy) for the GPU-vs-CPU comparison."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def usable_cores():
    n = len(os.sched_getaffinity(0))
    quota = None
    try:                                       # cgroup v2
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(p)
    except Exception:
        try:                                   # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except Exception:
            pass
    eff = n if quota is None else max(1, min(n, int(quota + 0.5)))
    return n, quota, eff


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=128)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--rhs", default="lin")
    ap.add_argument("--kappa", default="const")
    ap.add_argument("--variant", choices=["seq", "omp"], default="seq")
    ap.add_argument("--limit-s", type=float, default=10.0)
    ap.add_argument("--max-reps", type=int, default=8)
    ap.add_argument("--out-x", default=None)
    ap.add_argument("--config3", action="store_true",
                    help="BASELINE configs[2] shape: Q2, vertex-star patch smoother Richardson(PatchSolver,10,0.2) pre = post, FGMRES(5) "
                         "(test/LinearSolvers/GMGTests.jl:18-47,119-123) at --cells / --levels")
    args = ap.parse_args()
    naff, quota, eff = usable_cores()            # before the OpenMP runtime pins this thread

    import numpy as np
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    po = entry.import_package().poisson          # numpy-only input synthesis (no GPU, no torch)
    orc = entry.import_oracle()
    nc = (args.cells,) * 3
    if args.config3:
        return config3(args, po, orc, nc, naff, quota, eff)
    kap = po.smooth_kappa if args.kappa == "smooth" else None
    H = po.build_hierarchy(nc, args.levels, 1, kappa=kap)
    A0 = H["mats"][0]
    n = A0.shape[0]
    if args.rhs == "lin":
        b = po.dirichlet_lift_rhs(nc, 1)
        maxiter, atol, rtol = 20, 1e-14, 1e-6
    elif args.rhs == "manufactured":
        b = A0.matvec(po.nodal_values(nc, 1))
        maxiter, atol, rtol = 20, 1e-14, 1e-6
    else:
        b = po.random_rhs(n)
        maxiter, atol, rtol = 100, 1e-14, 1e-8
    orc.set_variant(args.variant)
    nthr = orc.threads()
    if args.variant == "omp":
        # parallel first touch of every operator array
        import ctypes as C
        L = orc.lib()

        def pcopy(a):
            out = np.empty_like(a)
            L.orc_parallel_copy(out.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p), C.c_int64(a.size), C.c_int(a.itemsize))
            return out
        for key in ("mats", "prolongations", "restrictions"):
            H[key] = [po.CSR(M.shape, pcopy(M.ptr), pcopy(M.idx), pcopy(M.val)) for M in H[key]]
        A0 = H["mats"][0]
        b = pcopy(b)
    t0 = time.perf_counter()
    g = orc.GMG(H["mats"], H["prolongations"], H["restrictions"], maxiter=1)
    t_setup = time.perf_counter() - t0
    reps, total = 0, 0.0
    while reps == 0 or (total < args.limit_s and reps < args.max_reps):
        t0 = time.perf_counter()
        x, nit, flag, hist = orc.cg_solve(A0, b, Pl=g, maxiter=maxiter, atol=atol, rtol=rtol)
        total += time.perf_counter() - t0
        reps += 1
    if args.out_x:
        np.save(args.out_x, x)
    print(json.dumps(dict(variant=args.variant, threads=int(nthr), seconds=total / reps, reps=reps, iters=int(nit), flag=int(flag),
                          hist=[float(v) for v in hist], setup_s=t_setup, dofs=int(n), affinity_cores=naff, cgroup_cpu_quota=quota,
                          usable_cores=eff, omp_env={k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OMP_PROC_BIND", "OMP_PLACES")})))


def config3(args, po, orc, nc, naff, quota, eff):
    """Q2 + vertex-star patch smoother + FGMRES(5): the oracle's patch solve (PatchSolvers.jl:279-300 restated) on a size it affords"""
    import numpy as np
    order = 2
    H = po.build_hierarchy(nc, args.levels, order)
    tabs = [po.vertex_star_patches(c, order) for c in H["ncells"][:-1]]
    b = po.dirichlet_lift_rhs(nc, order)
    orc.set_variant(args.variant)
    t0 = time.perf_counter()
    g = orc.GMG(H["mats"], H["prolongations"], H["restrictions"],
                pre_smoothers=[orc.Smoother(orc.PATCH, 10, 0.2, pp, pd) for pp, pd in tabs], maxiter=1)
    t_setup = time.perf_counter() - t0
    reps, total = 0, 0.0
    while reps == 0 or (total < args.limit_s and reps < args.max_reps):
        t0 = time.perf_counter()
        x, nit, flag, hist = orc.fgmres_solve(H["mats"][0], b, Pr=g, m=5, maxiter=20, atol=1e-14, rtol=1e-6)
        total += time.perf_counter() - t0
        reps += 1
    if args.out_x:
        np.save(args.out_x, x)
    print(json.dumps(dict(variant=args.variant, threads=int(orc.threads()), seconds=total / reps, reps=reps, iters=int(nit), flag=int(flag),
                          hist=[float(v) for v in hist], setup_s=t_setup, dofs=int(b.size), affinity_cores=naff, cgroup_cpu_quota=quota,
                          usable_cores=eff, omp_env={k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OMP_PROC_BIND", "OMP_PLACES")})))


if __name__ == "__main__":
    main()
