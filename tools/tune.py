"""Kernel-variant timing harness (GPU box): python tools_tune.py  -> one line per variant."""
import json, os, subprocess, sys
VARIANTS = json.loads(os.environ.get("TUNE_VARIANTS", "[]")) or [dict()]
KEYS = ("GMG_ONE_GATHER", "GMG_XCD_REMAP", "GMG_LANES_LOG2", "GMG_NT", "GMG_SELL", "GMG_SELL_UN", "GMG_SELL_BLOCK", "GMG_SELL_MAXPAD", "GMG_VDICT", "GMG_IDX16", "GMG_PATTERN", "GMG_PAT_UN", "GMG_PAT_WGS", "GMG_PAT_DBG", "GMG_PAT_SHARED", "GMG_PAT_RB", "GMG_PAT_DEFER", "GMG_PAT_DINV", "GMG_PAT_SMALL_WPB", "GMG_PAT_SMALL_WPB2", "GMG_PAT_EMIT")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np, torch
    import __graft_entry__ as entry
    pkg = entry.import_package(); po, S = pkg.poisson, pkg.solvers
    nc, nlev = (128,) * 3, 4
    H = po.build_hierarchy(nc, nlev, 1)
    b = po.dirichlet_lift_rhs(nc, 1)
    sm = [S.RichardsonSmoother(S.JacobiLinearSolver(), 10, 2.0 / 3.0)] * (nlev - 1)
    solver = S.CGSolver(S.GMGLinearSolver(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=sm, post_smoothers=sm, maxiter=1), maxiter=20, atol=1e-14, rtol=1e-6)
    for env in VARIANTS:
        for k in KEYS: os.environ.pop(k, None)
        for k, v in env.items(): os.environ[k] = str(v)
        ns = S.numerical_setup(S.symbolic_setup(solver, H["mats"][0]), H["mats"][0])
        bd = torch.from_numpy(b).cuda(); xd = torch.zeros_like(bd); torch.cuda.synchronize()
        import time
        for _ in range(2): xd.zero_(); torch.cuda.synchronize(); S.solve_(xd, ns, bd)
        ns.P_ns.profile(0, True)
        t0 = time.perf_counter()
        for _ in range(5): xd.zero_(); torch.cuda.synchronize(); S.solve_(xd, ns, bd)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        st = ns.P_ns.kernel_stats()
        ns.P_ns.profile(0, False)
        for _ in range(2): xd.zero_(); torch.cuda.synchronize(); S.solve_(xd, ns, bd)
        t0 = time.perf_counter()
        for _ in range(10): xd.zero_(); torch.cuda.synchronize(); S.solve_(xd, ns, bd)
        torch.cuda.synchronize(); dt_np = (time.perf_counter() - t0) / 10
        print(json.dumps(dict(env=env, ms_per_solve=round(dt * 1e3, 3), ms_noprof=round(dt_np * 1e3, 3), sweep_us=round(st["total_ms"] / st["launches"] * 1e3, 2),
                              sweep_GBs=round(st["alg_bytes"] / (st["total_ms"] / st["launches"] * 1e-3) / 1e9, 1), iters=solver.log.num_iters,
                              err=po.l2_error_sq(nc, 1, xd.cpu().numpy()))), flush=True)
        ns.P_ns.close(); del ns
else:
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "child"])
