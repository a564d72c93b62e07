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
