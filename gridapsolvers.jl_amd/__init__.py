"""gridapsolvers.jl_amd -- MI355X-native GMG V-cycle hot path of GridapSolvers.jl.

Contents (only what the hot path needs):
  csrc/      hand-written gfx950 HIP kernels + the C ABI (libgmgamd.so, include/gmg_amd.h)
  julia/     the Julia `ccall` binding that keeps the Gridap.Algebra surface
  abi.py     ctypes binding of the same C ABI
  solvers.py Python mirror of the reference's LinearSolver / setup / solve! interface
  poisson.py structured Q1/Q2 Poisson hierarchy generator (driver-side input synthesis)
  partition.py PartitionedArrays-style box row-partition for multi-GPU runs

The directory name contains a dot, so import it through
`__graft_entry__.import_package()` which registers it as `gridapsolvers_jl_amd`.
"""
from . import abi, build, poisson, solvers  # noqa: F401
from .solvers import *  # noqa: F401,F403
