#!/usr/bin/env python3
"""One CG iteration of a bench.py run out of a rocprofv3 --kernel-trace database: python tools/iteration_timeline.py <dir> [which]
The iterations are delimited by reduce_post_kernel (the norm of the residual posted to the host, CGSolvers.jl:111); `which` counts them
from the end of the trace (default: the last iteration in which no launch is bracketed by the bench's own HIP events).
Prints every kernel of the iteration (start offset us, duration, gap to the previous kernel's end, grid, name) and the sums by class."""
import collections, glob, os, sqlite3, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)
con = sqlite3.connect(f[0])
rows = con.execute("select start, end, grid_x, name from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "reduce_post_kernel" in r[3]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 12
# segments between two norms that hold a whole V-cycle (>= 20 sweep launches), counted from the end
segs = [(marks[i], marks[i + 1]) for i in range(len(marks) - 1) if sum("sweep_kernel" in r[3] for r in rows[marks[i] + 1:marks[i + 1] + 1]) >= 20]
if len(sys.argv) <= 2:
    # default: the last iteration whose launches are not bracketed by the bench's own HIP events (gmg_profile_enable: two bubbles of
    # ~5.7 us per sampled launch -- bench.py's per-level profiling runs at the end of the trace, every prof_stride-th finest sweep
    # of the timed solves): no idle gap above 2.5 us after the first launch
    for w in range(8, len(segs)):
        a_, b_ = segs[-w]
        seq = rows[a_ + 1:b_ + 1]
        if all(seq[i][0] - seq[i - 1][1] < 2500 for i in range(1, len(seq))):
            which = w
            break
a, b = segs[-which]
it = rows[a + 1:b + 1]
t0 = it[0][0]
prev = rows[a][1]
finest = max(r[2] for r in it if "sweep_kernel" in r[3])
cls = collections.OrderedDict()
def klass(g, n):
    if "sweep_kernel" in n and g == finest: return "finest-level sweeps"
    if "sells_smooth_kernel" in n: return "one-launch smoothing passes of levels >= 1 (grid %d)" % g
    if "sweep_kernel" in n: return "per-sweep launches of levels >= 1"
    if "dense_gemv" in n: return "coarse solve (dense inverse GEMV)"
    if any(k in n for k in ("cg_update", "xpby", "dot_partial", "reduce_", "post_scalar", "set_scalar")): return "CG's own vector kernels and reductions"
    if "r2mv_kernel<0" in n or "zsweep_kernel<1, true, false, 0>" in n: return "w = A p (+ first stage of dot(p, w))"
    if g == finest or g >= 200000: return "finest-level transfers and r -= A dx"
    return "transfers / residual updates of levels >= 1"
print(f"# {len(rows)} kernels in the trace; CG iteration = kernels {a + 1}..{b} ({len(it)} launches)")
for s, e, g, n in it:
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.2f} gap {(s - prev) / 1e3:7.2f} | {g:9d} | {n[:90]}")
    k = klass(g, n)
    c = cls.setdefault(k, [0, 0.0]); c[0] += 1; c[1] += (e - s) / 1e3
    prev = e
tot = sum(c[1] for c in cls.values())
wall = (it[-1][1] - rows[a][1]) / 1e3
print(f"# sum of kernel durations {tot:.1f} us in {wall:.1f} us of wall time (end of the previous norm to the end of this one)")
for k, c in cls.items():
    print(f"# {c[1]:8.1f} us {100 * c[1] / tot:5.1f} %  {c[0]:3d} launches  {k}")
