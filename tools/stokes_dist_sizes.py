import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_distributed as td, pathlib, tempfile
tmp = pathlib.Path(tempfile.mkdtemp())
for world, n, nlev, depth in [(4, 64, 4, 0), (4, 64, 4, 2), (8, 128, 5, 0)]:
    t0 = time.time()
    try:
        v = td._launch("gpu_stokes", world, (n, n), nlev, tmp, transport="host", timeout=800, extra_env={"GMG_TEST_DEPTH": str(depth)} if depth else None)
        print(world, n, nlev, depth, round(time.time() - t0, 1), {k: v[k] for k in ("iters", "iters_oracle", "gmg_iters", "gmg_iters_oracle", "rel_err", "true_residual", "exchanges", "overlap_levels")}, flush=True)
    except AssertionError as e:
        print(world, n, nlev, depth, "FAILED", str(e)[-800:], flush=True)
