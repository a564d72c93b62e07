"""Builds libgmgamd.so (the C-ABI shared library) in-tree with hipcc for gfx950."""
from __future__ import annotations

import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "gmg_amd.hip")
def deps():
    """every file compiled into libgmgamd.so: csrc/* (gmg_amd.hip includes kernels.hpp, comm.hpp, block.inc.hpp, ...) + the public header"""
    return sorted(glob.glob(os.path.join(HERE, "csrc", "*"))) + sorted(glob.glob(os.path.join(HERE, "..", "include", "*.h")))
LIB = os.path.join(HERE, "libgmgamd.so")


def lib_path():
    return LIB


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in deps())


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU)."""
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
           "-ffp-contract=off", "-o", LIB, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
