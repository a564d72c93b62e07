"""A/B of env switches on the 256^3 problem: python tools_tune256.py '<json list of env dicts>'"""
import json, os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import time, numpy as np, torch
    import __graft_entry__ as entry
    pkg = entry.import_package(); po, S = pkg.poisson, pkg.solvers
    cells, nlev = int(os.environ.get("TUNE_CELLS", "256")), int(os.environ.get("TUNE_LEVELS", "5"))
    nc = (cells,) * 3
    H = po.build_hierarchy(nc, nlev, 1); b = po.dirichlet_lift_rhs(nc, 1)
    sm = [S.RichardsonSmoother(S.JacobiLinearSolver(), 10, 2.0 / 3.0)] * (nlev - 1)
    solver = S.CGSolver(S.GMGLinearSolver(H["mats"], H["prolongations"], H["restrictions"], pre_smoothers=sm, post_smoothers=sm, maxiter=1), maxiter=20, atol=1e-14, rtol=1e-6)
    keys = set()
    for env in json.loads(os.environ["TUNE_VARIANTS"]): keys |= set(env)
    for env in json.loads(os.environ["TUNE_VARIANTS"]):
        for k in keys: os.environ.pop(k, None)
        for k, v in env.items(): os.environ[k] = str(v)
        ns = S.numerical_setup(S.symbolic_setup(solver, H["mats"][0]), H["mats"][0])
        bd = torch.from_numpy(b).cuda(); xd = torch.zeros_like(bd); torch.cuda.synchronize()
        for _ in range(2): xd.zero_(); torch.cuda.synchronize(); S.solve_(xd, ns, bd)
        ns.P_ns.profile(0, True); t0 = time.perf_counter()
        for _ in range(4): xd.zero_(); torch.cuda.synchronize(); S.solve_(xd, ns, bd)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4; st = ns.P_ns.kernel_stats()
        print(json.dumps(dict(env=env, ms_per_solve=round(dt * 1e3, 3), sweep_us=round(st["total_ms"] / st["launches"] * 1e3, 2), iters=solver.log.num_iters)), flush=True)
        ns.P_ns.close(); del ns
else:
    os.environ["TUNE_VARIANTS"] = sys.argv[1]
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "child"])
