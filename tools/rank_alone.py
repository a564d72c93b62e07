#!/usr/bin/env python3
"""The communication-free part of an N-rank iteration, measured: ONE rank of the N-rank partition runs alone on the GPU
(multigpu._AloneTransport: zero halos, no peers), so the kernels it launches -- own x own sweeps + boundary fix-ups on the finest level, the
overlapping levels with their redundant ghost rows, pack / unpack kernels, the replicated coarse levels -- are exactly those of that rank in
the real run, undisturbed.  Run under rocprofv3 --kernel-trace (tools/rank_alone.sh) and compare the kernel time per solve with the
single-GPU solve of the same per-rank problem (bench.py's weak anchor):  python tools/rank_alone.py [cells] [levels] [world] [rank]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
import importlib
pkg = entry.import_package()
mg = importlib.import_module(pkg.__name__ + ".multigpu")
pa = importlib.import_module(pkg.__name__ + ".partition")
import torch
cells, nlev, world, rank = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 288), (2, 6), (3, 8), (4, 0)))
rep_from, depths, table = mg.plan_partition(cells, nlev, world)
lengths = tuple(float(v) for v in pa.rank_grid(world, 3))
g = mg.DistributedGMG((cells,) * 3, nlev, rank, world, device_id=0, transport="alone", lengths=lengths, rep_from=rep_from, depth=depths, finest_depth=depths[0])
b = torch.from_numpy(g.rhs_lin()).cuda()
x = torch.zeros(g.n_own, dtype=torch.float64, device="cuda")
nsolve, iters = 4, 3
for k in range(nsolve + 1):
    x.zero_(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    log = g.cg_solve(b, x, maxiter=iters, atol=0.0, rtol=0.0)          # exactly `iters` iterations (the numbers mean nothing: no peers)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"rank {rank} of {world} alone: {cells}^3 cells per rank, {nlev} levels, depths {depths}, replicated from level {rep_from}: "
      f"{log.num_iters} iterations in {dt * 1e3:.3f} ms = {dt * 1e3 / max(log.num_iters, 1):.3f} ms per iteration (wall, host callbacks included); "
      f"halo exchanges per solve {g.comm_stats()[0] // (nsolve + 1)}")
print("finest sweep:", g.sweep_signature(0))
g.close()
