import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "child_process: the test body runs in a pytest child process of its own (>= 1e7-row problems, "
                                       "first in-process use of the partitioned path): a device fault there is ONE named failure, "
                                       "not the end of the whole record")


@pytest.hookimpl(tryfirst=True)
def pytest_pyfunc_call(pyfuncitem):
    """@pytest.mark.child_process: re-run exactly this test id in a fresh interpreter and report its outcome."""
    if pyfuncitem.get_closest_marker("child_process") is None or os.environ.get("GMG_TEST_IN_CHILD"):
        return None                                   # the normal in-process call
    import subprocess
    env = dict(os.environ, GMG_TEST_IN_CHILD="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "pytest", pyfuncitem.nodeid, "-x", "-q", "-p", "no:cacheprovider", "-m", "gpu or not gpu"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (out.stdout[-3000:] + "\n" + out.stderr[-2000:]).strip()
    if out.returncode == 5 or " skipped" in out.stdout and " passed" not in out.stdout:
        pytest.skip("child: " + tail[-300:])
    assert out.returncode == 0, f"child process of {pyfuncitem.nodeid} ended with rc {out.returncode}:\n{tail}"
    return True


# Crash diagnosability: every test announces itself BEFORE it runs, flushed, on the
# terminal and in a side file, so that a process abort (HSA runtime abort(), device fault) still names the
# test that was running.  pytest.ini turns the faulthandler plugin off: its all-threads dump used to fill
# the whole captured tail and push the runtime's own message out of it.
_START_LOG = os.environ.get("GMG_TEST_START_LOG", os.path.join(ROOT, "gpurun_out", "pytest_starts.log"))


def _announce(line):
    import time
    line = f"{line}  @{time.strftime('%H:%M:%S')}" + (" (child)" if os.environ.get("GMG_TEST_IN_CHILD") else "")
    try:
        sys.__stdout__.write("\n" + line + "\n")
        sys.__stdout__.flush()
    except Exception:
        pass
    try:
        os.makedirs(os.path.dirname(_START_LOG), exist_ok=True)
        with open(_START_LOG, "a") as f:
            f.write(line + "\n")
            f.flush()
            os.fsync(f.fileno())
    except Exception:
        pass


def pytest_runtest_logstart(nodeid, location):
    _announce(f"[start] {nodeid}")


def pytest_runtest_logfinish(nodeid, location):
    _announce(f"[done]  {nodeid}")


@pytest.fixture(scope="session")
def pkg():
    return entry.import_package()


@pytest.fixture(scope="session")
def orc():
    return entry.import_oracle()


@pytest.fixture(scope="session")
def po(pkg):
    return pkg.poisson


@pytest.fixture(scope="session")
def S(pkg):
    return pkg.solvers


_HCACHE = {}


@pytest.fixture(scope="session")
def hierarchy(po):
    def get(nc, nlev, order=1):
        key = (tuple(nc), nlev, order)
        if key not in _HCACHE:
            _HCACHE[key] = po.build_hierarchy(nc, nlev, order)
        return _HCACHE[key]
    return get


def rel_err(a, b):
    a = np.asarray(a); b = np.asarray(b)
    d = np.linalg.norm(a - b)
    n = np.linalg.norm(b)
    return d / n if n > 0 else d


def max_rel(a, b):
    """max|a-b| / max|b| -- the per-kernel parity measure."""
    a = np.asarray(a); b = np.asarray(b)
    m = np.max(np.abs(b)) if b.size else 0.0
    d = np.max(np.abs(a - b)) if b.size else 0.0
    return d / m if m > 0 else d
