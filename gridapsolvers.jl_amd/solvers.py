"""Host-side mirror of the reference's Gridap.Algebra surface for the GMG path.

The reference's host language (Julia) is not available in this image, so the
layer a GridapSolvers user touches is mirrored here in Python with the same
names, argument meaning, defaults and error behaviour, on top of the C ABI
(include/gmg_amd.h).  The Julia binding of the same ABI lives in
julia/GridapSolversAMD.jl.

    reference (src/LinearSolvers)                      here
    ------------------------------------------------   ---------------------------
    JacobiLinearSolver()            JacobiLinearSolvers.jl:6        JacobiLinearSolver
    RichardsonSmoother(M,niter,w)   RichardsonSmoothers.jl:21-30    RichardsonSmoother
    PatchSolver / BlockJacobiSolver PatchBasedSmoothers/*.jl        PatchSolver / BlockJacobiSolver
    GMGLinearSolver(mats,P,R;...)   GMGLinearSolvers.jl:48-69       GMGLinearSolver
    CGSolver(Pl;...)                Krylov/CGSolvers.jl:19          CGSolver
    FGMRESSolver(m,Pr;...)          Krylov/FGMRESSolvers.jl:26      FGMRESSolver
    BlockDiagonalSolver(blocks,solvers)   BlockSolvers/BlockDiagonalSolvers.jl:20-45     BlockDiagonalSolver
    BlockTriangularSolver(blocks,solvers,coeffs,half)  BlockTriangularSolvers.jl:55-85    BlockTriangularSolver
    LinearSystemBlock / MatrixBlock BlockSolvers/BlockSolverInterfaces.jl            same names
    symbolic_setup / numerical_setup / numerical_setup! / solve!    same names (solve_ = solve!)
    ConvergenceLog                  SolverInterfaces/ConvergenceLogs.jl:42   ConvergenceLog

All arithmetic happens in libgmgamd.so on the GPU.  Vectors may be numpy arrays
(host memory, copied in/out per call) or CUDA/HIP torch tensors (used in place).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import abi

__all__ = [
    "JacobiLinearSolver", "RichardsonSmoother", "PatchSolver", "BlockJacobiSolver", "LUSolver",
    "GMGLinearSolver", "CGSolver", "FGMRESSolver", "ConvergenceLog", "PatchProlongationOperator",
    "RichardsonLinearSolver", "BlockDiagonalSolver", "BlockTriangularSolver", "LinearSystemBlock", "MatrixBlock", "LinearSolverFromSmoother",
    "symbolic_setup", "numerical_setup", "numerical_setup_", "solve_", "mul_",
    "SOLVER_CONVERGED_ATOL", "SOLVER_CONVERGED_RTOL", "SOLVER_DIVERGED_MAXITER", "SOLVER_DIVERGED_BREAKDOWN",
]

SOLVER_CONVERGED_ATOL, SOLVER_CONVERGED_RTOL, SOLVER_DIVERGED_MAXITER, SOLVER_DIVERGED_BREAKDOWN = 0, 1, 2, 3


# ----------------------------------------------------------------------------
# small solver description objects (constructors only hold parameters)
# ----------------------------------------------------------------------------
class JacobiLinearSolver:
    """struct JacobiLinearSolver <: LinearSolver (JacobiLinearSolvers.jl:6)."""


class LUSolver:
    """Gridap.Algebra.LUSolver(): the default coarsest_solver (GMGLinearSolvers.jl:54)."""


class HostCallbackSolver:
    """coarsest_solver implemented by the host language: `fn(r) -> x` on numpy vectors (GMG_COARSE_HOST_CALLBACK).
    In Julia this is an `@cfunction` around `solve!(x, ns_coarse, r)` of any Gridap LinearSolver."""

    def __init__(self, fn):
        self.fn = fn


class PatchSolver:
    """PatchBasedSmoothers.PatchSolver restricted to what reaches solve!: the
    patch dof tables (patch_rows == patch_cols for :star assembly) -- PatchSolvers.jl:18-54."""
    kind = abi.PATCH_LU

    def __init__(self, patch_ptr, patch_dofs, patch_cols=None, patch_mats=None, factors=None, pivots=None):
        """patch_dofs = patch_rows.  Optional, as held by the reference's PatchNS (PatchSolvers.jl:100-150):
        patch_cols (separate column table), patch_mats (the solver's own assembled patch matrices, column-major,
        concatenated) or factors (+ 1-based LAPACK pivots) = the output of lu!(patch_mat)."""
        self.patch_ptr = np.ascontiguousarray(patch_ptr, dtype=np.int64)
        self.patch_dofs = np.ascontiguousarray(patch_dofs, dtype=np.int32)
        if self.patch_ptr.ndim != 1 or self.patch_ptr.size < 1:
            raise ValueError("patch_ptr must have npatch+1 entries")
        self.patch_cols = None if patch_cols is None else np.ascontiguousarray(patch_cols, dtype=np.int32)
        if patch_mats is not None and factors is not None:
            raise ValueError("give patch_mats or factors, not both")
        self.patch_mats = None if patch_mats is None else np.ascontiguousarray(patch_mats, dtype=np.float64)
        self.factors = None if factors is None else np.ascontiguousarray(factors, dtype=np.float64)
        self.pivots = None if pivots is None else np.ascontiguousarray(pivots, dtype=np.int32)


class BlockJacobiSolver(PatchSolver):
    """BlockJacobiSolver(patch_rows, patch_cols) -- BlockJacobiSolvers.jl:2-16 (NoPivot LU, :162)."""
    kind = abi.PATCH_NOPIVOT


class PatchProlongationOperator:
    """PatchProlongationOperator (PatchTransferOperators.jl:2-60,153-172) reduced to what its mul! needs: the
    plain prolongation matrix P and the patch dof tables; y = P x - sum_p A_pp^-1 (A P x)_p."""

    def __init__(self, P, patch_ptr, patch_dofs, pivoting=True, rhs=None):
        """rhs: assembled rhs form of the local problems when it is not the level operator (StokesGMG.jl:125-127 passes graddiv)."""
        self.P = P
        self.rhs = rhs
        self.patch_ptr = np.ascontiguousarray(patch_ptr, dtype=np.int64)
        self.patch_dofs = np.ascontiguousarray(patch_dofs, dtype=np.int64)
        self.kind = abi.PATCH_LU if pivoting else abi.PATCH_NOPIVOT


class RichardsonSmoother:
    """RichardsonSmoother(M, niter=1, omega=1.0) -- RichardsonSmoothers.jl:21-30."""

    def __init__(self, M, niter=1, omega=1.0):
        if not isinstance(M, (JacobiLinearSolver, PatchSolver)):
            raise TypeError("RichardsonSmoother: M must be JacobiLinearSolver, PatchSolver or BlockJacobiSolver")
        self.M, self.niter, self.omega = M, int(niter), float(omega)


class LinearSolverFromSmoother:
    """LinearSolverFromSmoother(smoother) -- LinearSolverFromSmoothers.jl:2-4: x = 0; r = copy(b); solve!(x,smoother,r)."""

    def __init__(self, smoother):
        if not isinstance(smoother, RichardsonSmoother):
            raise TypeError("smoother must be a RichardsonSmoother")
        self.smoother = smoother


class ConvergenceLog:
    """ConvergenceLog{Float64}: name, tols, num_iters, residuals (ConvergenceLogs.jl:42-60)."""

    def __init__(self, name, maxiter, atol, rtol):
        self.name, self.maxiter, self.atol, self.rtol = name, int(maxiter), float(atol), float(rtol)
        self.num_iters = 0
        self.residuals = np.zeros(self.maxiter + 1)
        self.flag = None

    def _fill(self, res, hist):
        self.num_iters = int(res.niters)
        self.residuals[:] = 0.0
        self.residuals[: self.num_iters + 1] = hist[: self.num_iters + 1]
        self.flag = int(res.flag)

    def summary(self):
        r = self.residuals[self.num_iters]
        return (f"Convergence[{self.name}]: conv_flag={self.flag}, niter={self.num_iters}, "
                f"r_abs={r}, r_rel={r / self.residuals[0] if self.residuals[0] else float('nan')}")


_MODES = {"preconditioner": abi.MODE_PRECONDITIONER, "solver": abi.MODE_SOLVER}
_CYCLES = {"v_cycle": abi.V_CYCLE, "w_cycle": abi.W_CYCLE, "f_cycle": abi.F_CYCLE}


class GMGLinearSolver:
    """GMGLinearSolver(smatrices, interp, restrict; pre_smoothers, post_smoothers,
    coarsest_solver, mode, cycle_type, maxiter, atol, rtol) -- GMGLinearSolvers.jl:48-69.

    smatrices / interp / restrict are sequences of CSR-like objects with fields
    (shape, ptr, idx, val), 0-based (e.g. poisson.CSR or scipy.sparse.csr_matrix
    via `from_scipy`).  restrict=None means R = P^T (the :residual-mode transfer of
    GridTransferOperators.jl:202-209 on nested meshes)."""

    def __init__(self, smatrices, interp, restrict=None, pre_smoothers=None, post_smoothers=None,
                 coarsest_solver=None, mode="preconditioner", cycle_type="v_cycle",
                 maxiter=100, atol=1.0e-14, rtol=1.0e-8, verbose=False, options=None, pin_vectors=False):
        nlev = len(smatrices)
        if pre_smoothers is None:  # Fill(RichardsonSmoother(JacobiLinearSolver(),10),nlev-1)
            pre_smoothers = [RichardsonSmoother(JacobiLinearSolver(), 10) for _ in range(nlev - 1)]
        if post_smoothers is None:
            post_smoothers = pre_smoothers
        if restrict is None:
            restrict = [None] * (nlev - 1)
        # @check length(smatrices)-1 == length(interp) == ... (GMGLinearSolvers.jl:59)
        if not (nlev - 1 == len(interp) == len(restrict) == len(pre_smoothers) == len(post_smoothers)):
            raise ValueError("length(smatrices)-1 must equal the number of transfer operators and smoothers")
        if mode not in _MODES:          # :60
            raise ValueError("mode must be 'preconditioner' or 'solver'")
        if cycle_type not in _CYCLES:   # :61
            raise ValueError("cycle_type must be 'v_cycle', 'w_cycle' or 'f_cycle'")
        # coarsest_solver: LUSolver() (default, GMGLinearSolvers.jl:54), CGSolver(JacobiLinearSolver();...) on the device, or any
        # host-side solver wrapped in HostCallbackSolver (the analogue of passing PETSc / UMFPACK objects in Julia)
        if coarsest_solver is not None and not isinstance(coarsest_solver, (LUSolver, HostCallbackSolver)):
            if not (isinstance(coarsest_solver, CGSolver) and isinstance(coarsest_solver.Pl, JacobiLinearSolver)
                    and not coarsest_solver.flexible):
                raise NotImplementedError("coarsest_solver on the device: LUSolver(), CGSolver(JacobiLinearSolver()) or HostCallbackSolver(fn)")
        self.coarsest_solver = coarsest_solver
        self.smatrices, self.interp, self.restrict = list(smatrices), list(interp), list(restrict)
        self.pre_smoothers, self.post_smoothers = list(pre_smoothers), list(post_smoothers)
        self.mode, self.cycle_type = mode, cycle_type
        self.verbose = int(verbose)
        self.log = ConvergenceLog("GMG", maxiter, atol, rtol)
        # device-side policy (no reference counterpart): `options` = {key: number} handed to gmg_set_option before the operators are
        # set (storage layout, sweep kernels, one-launch passes, x0_zero, ...; keys in include/gmg_amd.h); `pin_vectors` = the
        # numerical setup page-locks the host vectors it is called with, once (the pattern of ext/GridapPETScExt/PETScCaches.jl:23-36)
        self.options = dict(options or {})
        self.pin_vectors = bool(pin_vectors)

    def num_levels(self):
        return len(self.smatrices)


class CGSolver:
    """CGSolver(Pl; maxiter=1000, atol=1e-12, rtol=1e-6, flexible=false) -- CGSolvers.jl:19."""

    def __init__(self, Pl=None, maxiter=1000, atol=1e-12, rtol=1.0e-6, flexible=False, verbose=0, name="CG"):
        self.Pl, self.flexible = Pl, bool(flexible)
        self.log = ConvergenceLog(name, maxiter, atol, rtol)


class RichardsonLinearSolver:
    """RichardsonLinearSolver(omega, maxiter; Pl=nothing, rtol=1e-10, atol=1e-6) -- RichardsonLinearSolvers.jl:13-23
    (scalar omega on the device path)."""

    def __init__(self, omega, maxiter, Pl=None, rtol=1e-10, atol=1e-6, verbose=True, name="RichardsonLinearSolver"):
        if not np.isscalar(omega):
            raise NotImplementedError("vector-valued omega is not on the device path")
        self.omega, self.Pl = float(omega), Pl
        self.log = ConvergenceLog(name, maxiter, atol, rtol)


class FGMRESSolver:
    """FGMRESSolver(m, Pr; Pl=nothing, restart=false, m_add=1, maxiter=100, atol=1e-12,
    rtol=1e-6) -- FGMRESSolvers.jl:26."""

    def __init__(self, m, Pr, Pl=None, restart=False, m_add=1, maxiter=100, atol=1e-12, rtol=1.0e-6,
                 verbose=False, name="FGMRES"):
        # Pl (KrylovUtils.jl:14-18,46-50): None | JacobiLinearSolver() | LinearSolverFromSmoother(finest pre-smoother) on the
        # matrix of Pr's handle; the GMG itself serves on one side only
        if Pl is not None and not isinstance(Pl, (JacobiLinearSolver, LinearSolverFromSmoother)):
            raise NotImplementedError("FGMRES Pl on the device: JacobiLinearSolver() or LinearSolverFromSmoother(...)")
        self.Pl = Pl
        self.m, self.Pr, self.restart, self.m_add = int(m), Pr, bool(restart), int(m_add)
        self.log = ConvergenceLog(name, maxiter, atol, rtol)


class LinearSystemBlock:
    """LinearSystemBlock(): the block is taken from the system matrix (BlockSolverInterfaces.jl)."""


class MatrixBlock:
    """MatrixBlock(mat): an explicit matrix (also what a BiformBlock assembles to)."""

    def __init__(self, mat):
        self.mat = mat


class BlockDiagonalSolver:
    """BlockDiagonalSolver(blocks, solvers) / BlockDiagonalSolver(solvers) -- BlockDiagonalSolvers.jl:20-45.
    solvers[i]: GMGLinearSolver | CGSolver(JacobiLinearSolver()) | LUSolver() | JacobiLinearSolver()."""

    half = "diagonal"

    def __init__(self, blocks, solvers=None):
        if solvers is None:
            blocks, solvers = None, blocks
        self.solvers = list(solvers)
        nb = len(self.solvers)
        diag = [LinearSystemBlock() for _ in range(nb)] if blocks is None else list(blocks)
        if len(diag) != nb:
            raise ValueError("blocks and solvers must have the same length")   # @check :27
        self.blocks = [[diag[i] if i == j else LinearSystemBlock() for j in range(nb)] for i in range(nb)]
        self.coeffs = np.ones((nb, nb))


class BlockTriangularSolver:
    """BlockTriangularSolver(blocks, solvers, coeffs=fill(1.0,size(blocks)), half=:upper) /
    BlockTriangularSolver(solvers; coeffs, half) -- BlockTriangularSolvers.jl:55-85."""

    def __init__(self, blocks, solvers=None, coeffs=None, half="upper"):
        if solvers is None:
            blocks, solvers = None, blocks
        self.solvers = list(solvers)
        nb = len(self.solvers)
        if blocks is None:
            blocks = [[LinearSystemBlock() for _ in range(nb)] for _ in range(nb)]
        self.blocks = [list(row) for row in blocks]
        if len(self.blocks) != nb or any(len(r) != nb for r in self.blocks):
            raise ValueError("blocks must be a square matrix matching solvers")  # @check :63-64
        if half not in ("upper", "lower"):
            raise ValueError("half must be :upper or :lower")                     # @check :65
        self.coeffs = np.ones((nb, nb)) if coeffs is None else np.asarray(coeffs, dtype=np.float64)
        if self.coeffs.shape != (nb, nb):
            raise ValueError("coeffs must match blocks")
        self.half = half


# ----------------------------------------------------------------------------
# vector / matrix marshalling
# ----------------------------------------------------------------------------
def _is_device(v):
    return hasattr(v, "data_ptr") and getattr(v, "is_cuda", False)


def _vec(v, n=None, writable=False):
    """-> (pointer, memspace, keepalive)."""
    if _is_device(v):
        import torch
        if v.dtype != torch.float64 or not v.is_contiguous():
            raise TypeError("device vectors must be contiguous float64 tensors")
        if n is not None and v.numel() != n:
            raise ValueError(f"vector length {v.numel()} != {n}")
        return C.c_void_p(v.data_ptr()), abi.MEM_DEVICE, v
    if not isinstance(v, np.ndarray) or v.dtype != np.float64 or not v.flags.c_contiguous:
        if writable:
            raise TypeError("output vectors must be contiguous float64 numpy arrays (or CUDA tensors)")
        v = np.ascontiguousarray(v, dtype=np.float64)
    if n is not None and v.size != n:
        raise ValueError(f"vector length {v.size} != {n}")
    return C.c_void_p(v.ctypes.data), abi.MEM_HOST, v


def _csr_fields(M):
    """Accept poisson.CSR, scipy.sparse CSR/CSC, or any object with shape/ptr/idx/val."""
    if hasattr(M, "indptr"):  # scipy
        layout = abi.CSC if M.format == "csc" else abi.CSR
        if M.format not in ("csr", "csc"):
            M = M.tocsr(); layout = abi.CSR
        return M.shape, np.ascontiguousarray(M.indptr), np.ascontiguousarray(M.indices), \
            np.ascontiguousarray(M.data, dtype=np.float64), layout, 0
    base = getattr(M, "index_base", 0)
    layout = getattr(M, "layout", abi.CSR)
    return M.shape, np.ascontiguousarray(M.ptr), np.ascontiguousarray(M.idx), \
        np.ascontiguousarray(M.val, dtype=np.float64), layout, base


def _set_op(fn, h, lev, M):
    if hasattr(M, "row_blocks"):
        # an operator delivered as consecutive row blocks (poisson.StreamedCSR): gmg_set_operator_rows per block, the block
        # is dropped right after -- the host never holds the whole CSR
        lib = abi.load()
        op = {"gmg_set_matrix": abi.OP_A, "gmg_set_prolongation": abi.OP_P, "gmg_set_restriction": abi.OP_R}[fn.__name__]
        seen = 0
        plan = M.row_plan() if (hasattr(M, "row_plan") and os.environ.get("GMG_STREAM_REPEAT", "1") != "0") else (("block", r0, B) for r0, B in M.row_blocks())
        for item in plan:
            if item[0] == "repeat":                         # the last nrows_block rows recur: no arrays, nothing hashed
                _, nrb, count, cshift = item
                abi.check(h, lib.gmg_set_operator_rows_repeat(h, lev, op, nrb, count, cshift))
                seen += nrb * count
                continue
            _, row0, B = item
            if row0 != seen:
                raise ValueError("row-block stream out of order")
            ptr = np.ascontiguousarray(B.ptr, dtype=np.int64)
            idx = np.ascontiguousarray(B.idx, dtype=np.int64)
            val = np.ascontiguousarray(B.val, dtype=np.float64)
            abi.check(h, lib.gmg_set_operator_rows(h, lev, op, M.shape[0], M.shape[1], row0, B.shape[0], C.c_void_p(ptr.ctypes.data),
                                                   C.c_void_p(idx.ctypes.data), C.c_void_p(val.ctypes.data), 0, 8))
            seen = row0 + B.shape[0]
        if seen != M.shape[0]:
            raise ValueError("row-block stream ended early")
        return
    shape, ptr, idx, val, layout, base = _csr_fields(M)
    if ptr.dtype != idx.dtype:
        if idx.dtype == np.int32 and val.size < 2 ** 31 - 1:   # narrow the (short) pointer array instead of widening the index array
            ptr = ptr.astype(np.int32)
        else:
            ptr = ptr.astype(np.int64); idx = idx.astype(np.int64)
    if ptr.dtype not in (np.int32, np.int64):
        raise TypeError("index arrays must be int32 or int64")
    nnz = int(val.size)
    abi.check(h, fn(h, lev, shape[0], shape[1], nnz, C.c_void_p(ptr.ctypes.data), C.c_void_p(idx.ctypes.data),
                    C.c_void_p(val.ctypes.data), layout, base, ptr.dtype.itemsize))


# ----------------------------------------------------------------------------
# setups
# ----------------------------------------------------------------------------
class GMGSymbolicSetup:
    """GMGLinearSolvers.jl:160-170: no work."""

    def __init__(self, solver):
        self.solver = solver


class GMGNumericalSetup:
    """GMGNumericalSetup (GMGLinearSolvers.jl:172-210) backed by a native handle;
    released by a finalizer like ext/PardisoExt.jl:54-61."""

    def __init__(self, solver, mat, device_id=None):
        lib = abi.load()
        self.solver = solver
        self._lib = lib
        if device_id is None:
            device_id = 0
            try:
                import torch
                if torch.cuda.is_available():
                    device_id = torch.cuda.current_device()
            except Exception:
                pass
        h = C.c_void_p()
        abi.check(None, lib.gmg_create(C.byref(h), solver.num_levels(), device_id))
        self.h = h
        self._pinned = []          # host arrays registered with the handle, kept alive while they are (LRU, at most 8)
        try:
            for k, v in solver.options.items():
                self.set_option(k, v)
            self._upload(mat)
        except Exception:
            self.close()
            raise

    # -- per-handle policy (gmg_set_option) and page-locked host vectors (gmg_host_register)
    def set_option(self, key, value):
        abi.check(self.h, self._lib.gmg_set_option(self.h, str(key).encode(), float(value)))

    def get_option(self, key):
        """-> (effective value or None when the built-in default applies, source: 'default' | 'handle' | 'environment')"""
        v, src = C.c_double(), C.c_int()
        abi.check(self.h, self._lib.gmg_get_option(self.h, str(key).encode(), C.byref(v), C.byref(src)))
        return (None if v.value != v.value else v.value), ("default", "handle", "environment")[src.value]

    def set_stream(self, stream=None):
        """gmg_set_stream: issue the handle's work on the caller's HIP stream (an integer hipStream_t, a torch.cuda.Stream, or None for
        the handle's own) -- ordered with the caller's kernels on that stream, no device synchronisation in between."""
        abi.check(self.h, self._lib.gmg_set_stream(self.h, abi.stream_arg(stream)))

    def get_stream(self):
        out = C.c_void_p()
        abi.check(self.h, self._lib.gmg_get_stream(self.h, C.byref(out)))
        return out.value or 0

    def setup(self):
        """gmg_setup again (after set_option changed a layout option)."""
        abi.check(self.h, self._lib.gmg_setup(self.h))

    def register_host(self, arr):
        abi.check(self.h, self._lib.gmg_host_register(self.h, C.c_void_p(arr.ctypes.data), arr.nbytes))

    def unregister_host(self, arr):
        abi.check(self.h, self._lib.gmg_host_unregister(self.h, C.c_void_p(arr.ctypes.data)))

    def pin(self, *arrays):
        """Page-lock host vectors for the life of this setup (or until 8 newer ones displaced them); holds a reference to each.
        Keyed by address range, as the C registry is: two views of one buffer are one registration."""
        for a in arrays:
            if not isinstance(a, np.ndarray) or a.nbytes == 0:
                continue
            key = (a.ctypes.data, a.nbytes)
            hit = next((i for i, (k, _q) in enumerate(self._pinned) if k == key), None)
            if hit is not None:
                self._pinned.append(self._pinned.pop(hit))          # most recently used last
                continue
            if len(self._pinned) >= 8:
                _k, old = self._pinned.pop(0)
                try:
                    self.unregister_host(old)
                except abi.GmgError:                                 # (a range another view had already released)
                    pass
            self.register_host(a)
            self._pinned.append((key, a))

    def host_io_stats(self):
        up, down, nreg = C.c_int64(), C.c_int64(), C.c_int64()
        abi.check(self.h, self._lib.gmg_get_host_io_stats(self.h, C.byref(up), C.byref(down), C.byref(nreg)))
        return dict(bytes_up=up.value, bytes_down=down.value, registered=nreg.value)

    def persist_retries(self):
        r, a = C.c_int64(), C.c_int()
        abi.check(self.h, self._lib.gmg_get_persist_retries(self.h, C.byref(r), C.byref(a)))
        return dict(retries=r.value, persist_active=bool(a.value))

    def _upload(self, mat):
        lib, h, s = self._lib, self.h, self.solver
        nlev = s.num_levels()
        mats = list(s.smatrices)
        if mat is not None:
            mats[0] = mat  # smatrices[1] = mat, GMGLinearSolvers.jl:338
        for l, A in enumerate(mats):
            _set_op(lib.gmg_set_matrix, h, l, A)
        for l in range(nlev - 1):
            ip = s.interp[l]
            if isinstance(ip, PatchProlongationOperator):
                _set_op(lib.gmg_set_prolongation, h, l, ip.P)
                abi.check(h, lib.gmg_set_prolongation_patch_correction(
                    h, l, ip.kind, ip.patch_ptr.size - 1, C.c_void_p(ip.patch_ptr.ctypes.data),
                    C.c_void_p(ip.patch_dofs.ctypes.data), 0, 8))
                if ip.rhs is not None:
                    shape, ptr, idx, val, layout, base = _csr_fields(ip.rhs)
                    if ptr.dtype != idx.dtype:
                        ptr = ptr.astype(np.int64); idx = idx.astype(np.int64)
                    abi.check(h, lib.gmg_set_prolongation_patch_correction_rhs(
                        h, l, shape[0], int(val.size), C.c_void_p(ptr.ctypes.data), C.c_void_p(idx.ctypes.data),
                        C.c_void_p(val.ctypes.data), layout, base, ptr.dtype.itemsize))
            else:
                _set_op(lib.gmg_set_prolongation, h, l, ip)
            if s.restrict[l] is not None:
                _set_op(lib.gmg_set_restriction, h, l, s.restrict[l])
            pre, post = s.pre_smoothers[l], s.post_smoothers[l]
            if post is pre:
                self._set_smoother(l, abi.PRE_AND_POST, pre)
            else:
                self._set_smoother(l, abi.PRE, pre)
                self._set_smoother(l, abi.POST, post)
        abi.check(h, lib.gmg_set_options(h, _MODES[s.mode], _CYCLES[s.cycle_type], s.log.maxiter, s.log.atol, s.log.rtol))
        abi.check(h, lib.gmg_set_verbose(h, s.verbose))
        cs = s.coarsest_solver
        if isinstance(cs, CGSolver):
            abi.check(h, lib.gmg_set_coarse_solver(h, abi.COARSE_CG_JACOBI, cs.log.maxiter, cs.log.atol, cs.log.rtol, None, None))
        elif isinstance(cs, HostCallbackSolver):
            def _cb(ctx, n, r, x, fn=cs.fn):
                try:
                    out = np.asarray(fn(np.ctypeslib.as_array(r, shape=(n,)).copy()), dtype=np.float64)
                    np.ctypeslib.as_array(x, shape=(n,))[:] = out
                    return 0
                except Exception:      # never unwind through the C frames
                    return 1
            self._coarse_cb = abi.COARSE_SOLVE_FN(_cb)
            abi.check(h, lib.gmg_set_coarse_solver(h, abi.COARSE_HOST_CALLBACK, 0, 0.0, 0.0, C.cast(self._coarse_cb, C.c_void_p), None))
        abi.check(h, lib.gmg_setup(h))
        self.n = int(mats[0].shape[0])
        self.sizes = [int(A.shape[0]) for A in mats]

    def _set_smoother(self, l, which, sm):
        lib, h = self._lib, self.h
        if not isinstance(sm, RichardsonSmoother):
            raise TypeError("smoothers must be RichardsonSmoother objects")
        if isinstance(sm.M, JacobiLinearSolver):
            abi.check(h, lib.gmg_set_smoother_jacobi(h, l, which, sm.niter, sm.omega))
        else:
            M = sm.M
            pp = M.patch_ptr
            if M.patch_cols is None and M.patch_mats is None and M.factors is None:
                if M.patch_dofs.dtype == np.int32 and int(pp[-1]) < 2 ** 31 - 1:
                    # int32 tables as they are (1.7e7 vertex-star patches of a 256^3 Q2 level: 4.6e8 entries), the short pointer array narrowed
                    pp32, pd = np.ascontiguousarray(pp, dtype=np.int32), np.ascontiguousarray(M.patch_dofs)
                    abi.check(h, lib.gmg_set_smoother_patch(h, l, which, sm.niter, sm.omega, M.kind, pp.size - 1,
                                                            C.c_void_p(pp32.ctypes.data), C.c_void_p(pd.ctypes.data), 0, 4))
                    return
            pp, pd = np.ascontiguousarray(pp, dtype=np.int64), M.patch_dofs.astype(np.int64)
            if M.patch_cols is None and M.patch_mats is None and M.factors is None:
                abi.check(h, lib.gmg_set_smoother_patch(h, l, which, sm.niter, sm.omega, M.kind, pp.size - 1,
                                                        C.c_void_p(pp.ctypes.data), C.c_void_p(pd.ctypes.data), 0, 8))
            else:
                pc = None if M.patch_cols is None else M.patch_cols.astype(np.int64)
                blk = M.patch_mats if M.patch_mats is not None else M.factors
                abi.check(h, lib.gmg_set_smoother_patch_matrices(
                    h, l, which, sm.niter, sm.omega, M.kind, pp.size - 1, C.c_void_p(pp.ctypes.data), C.c_void_p(pd.ctypes.data),
                    None if pc is None else C.c_void_p(pc.ctypes.data), 0, 8,
                    None if blk is None else C.c_void_p(blk.ctypes.data), 1 if M.factors is not None else 0,
                    None if M.pivots is None else C.c_void_p(M.pivots.ctypes.data)))

    # -- numerical_setup!(ns, A): FromMatrices variant is unsupported in the reference
    #    (GMGLinearSolvers.jl:249-258 logs @error); here new values on the same pattern are accepted.
    def update(self, mat, smatrices=None):
        """numerical_setup!(ns, A): new values on the same sparsity.  `smatrices` (optional) = the re-assembled matrices of ALL
        levels, as the FromWeakform variant recomputes them (GMGLinearSolvers.jl:260-297); default: only the finest changes."""
        mats = [mat] if smatrices is None else [mat] + list(smatrices[1:])
        for l, M in enumerate(mats):
            if M is None:
                continue
            shape, ptr, idx, val, layout, base = _csr_fields(M)
            if layout == abi.CSC:           # values in the caller's CSC order (the level was set in CSC layout): scattered by the library
                st = self._lib.gmg_update_values_csc(self.h, l, C.c_void_p(ptr.ctypes.data), C.c_void_p(idx.ctypes.data),
                                                     C.c_void_p(val.ctypes.data), base, ptr.dtype.itemsize)
            else:
                st = self._lib.gmg_update_values(self.h, l, C.c_void_p(val.ctypes.data))
            if st == abi.ERR_UNSUPPORTED:       # level held in row-pattern form only: hand the whole matrix over again
                _set_op(self._lib.gmg_set_matrix, self.h, l, M)
            else:
                abi.check(self.h, st)
        abi.check(self.h, self._lib.gmg_setup(self.h))
        return self

    def close(self):
        if getattr(self, "h", None):
            self._lib.gmg_destroy(self.h)      # unregisters what is still page-locked (never frees caller memory)
            self.h = None
            self._pinned = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- operator-level helpers (duck-typed mul! / smoother solve!)
    def op_apply(self, lev, op, x, y):
        px, ms, _k1 = _vec(x)
        py, ms2, _k2 = _vec(y, writable=True)
        if ms != ms2:
            raise TypeError("x and y must live in the same memory space")
        abi.check(self.h, self._lib.gmg_op_apply(self.h, lev, op, px, py, ms))
        return y

    def smooth(self, lev, x, r, which=abi.PRE):
        px, ms, _k1 = _vec(x, self.sizes[lev], writable=True)
        pr, ms2, _k2 = _vec(r, self.sizes[lev], writable=True)
        if ms != ms2:
            raise TypeError("x and r must live in the same memory space")
        abi.check(self.h, self._lib.gmg_smooth(self.h, lev, which, px, pr, ms))
        return x, r

    def precond(self, lev, r, dx, which=abi.PRE):
        pr, ms, _k1 = _vec(r, self.sizes[lev])
        pd, ms2, _k2 = _vec(dx, self.sizes[lev], writable=True)
        if ms != ms2:
            raise TypeError("r and dx must live in the same memory space")
        abi.check(self.h, self._lib.gmg_precond_apply(self.h, lev, which, pr, pd, ms))
        return dx

    def coarse_solve(self, r, x):
        pr, ms, _k1 = _vec(r, self.sizes[-1])
        px, ms2, _k2 = _vec(x, self.sizes[-1], writable=True)
        if ms != ms2:
            raise TypeError("r and x must live in the same memory space")
        abi.check(self.h, self._lib.gmg_coarse_solve(self.h, pr, px, ms))
        return x

    def dot(self, a, b):
        pa, ms, _k1 = _vec(a)
        pb, ms2, _k2 = _vec(b)
        out = C.c_double(0.0)
        n = a.numel() if _is_device(a) else _k1.size
        abi.check(self.h, self._lib.gmg_dot(self.h, n, pa, pb, ms, C.byref(out)))
        return out.value

    def coarse_log(self):
        res = abi.Result()
        abi.check(self.h, self._lib.gmg_get_coarse_log(self.h, C.byref(res)))
        return dict(niters=int(res.niters), flag=int(res.flag), res0=res.res0, res=res.res)

    def fill_log(self):
        """ns.solver.log of the GMG after it ran inside a Krylov call (gmg_get_log)."""
        log = self.solver.log
        res = abi.Result()
        hist = np.zeros(log.maxiter + 1)
        abi.check(self.h, self._lib.gmg_get_log(self.h, C.byref(res), C.c_void_p(hist.ctypes.data), hist.size))
        log._fill(res, hist)
        return log

    def profile(self, lev=0, enable=True):
        abi.check(self.h, self._lib.gmg_profile_enable(self.h, lev, 1 if enable else 0))

    def kernel_stats(self):
        st = abi.KernelStats()
        abi.check(self.h, self._lib.gmg_get_kernel_stats(self.h, C.byref(st)))
        ms, nl, lb = (C.c_double * 3)(), (C.c_int64 * 3)(), (C.c_double * 3)()
        abi.check(self.h, self._lib.gmg_get_kernel_stats_by_variant(self.h, ms, nl, lb))
        names = ("x_every_sweep", "x_untouched", "x_two_increments")
        by_variant = {names[v]: dict(launches=int(nl[v]), total_ms=float(ms[v]), avg_ms=float(ms[v]) / nl[v], layout_bytes=float(lb[v]))
                      for v in range(3) if nl[v]}
        return dict(launches=st.launches, total_ms=st.total_ms, alg_bytes=st.alg_bytes, rows=st.rows, nnz=st.nnz,
                    layout_bytes=st.layout_bytes, fused_passes=st.fused_passes, by_variant=by_variant)

    def stream_probe(self, nbytes=1 << 30, reps=10):
        v = C.c_double(0.0)
        abi.check(self.h, self._lib.gmg_stream_probe(self.h, nbytes, reps, C.byref(v)))
        return v.value

    def stream_probe_read(self, nbytes=1 << 30, reps=10):
        v = C.c_double(0.0)
        abi.check(self.h, self._lib.gmg_stream_probe_read(self.h, nbytes, reps, C.byref(v)))
        return v.value

    def model_bytes(self):
        a, b = C.c_double(0.0), C.c_double(0.0)
        abi.check(self.h, self._lib.gmg_model_bytes(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def level_format(self, lev=0):
        a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
        d, e = C.c_double(0.0), C.c_double(0.0)
        abi.check(self.h, self._lib.gmg_level_format(self.h, lev, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(e)))
        return dict(layout=("CSR-stream", "SELL-64", "SELL-P", "SELL-O")[a.value], row_patterns=(a.value == 2),
                    value_dictionary=bool(b.value), idx16=bool(c.value), stream_bytes_per_nnz=d.value, padding=e.value)

    def sweep_signature(self, lev=0):
        buf = C.create_string_buffer(128)
        abi.check(self.h, self._lib.gmg_sweep_signature(self.h, lev, buf, 128))
        return buf.value.decode()

    def device_bytes(self):
        v = C.c_int64(0)
        abi.check(self.h, self._lib.gmg_device_bytes(self.h, C.byref(v)))
        return v.value


def _set_block(fn, h, i, j, M):
    shape, ptr, idx, val, layout, base = _csr_fields(M)
    if ptr.dtype != idx.dtype:
        if idx.dtype == np.int32 and val.size < 2 ** 31 - 1:   # narrow the (short) pointer array instead of widening the index array
            ptr = ptr.astype(np.int32)
        else:
            ptr = ptr.astype(np.int64); idx = idx.astype(np.int64)
    abi.check_block(h, fn(h, i, j, shape[0], shape[1], int(val.size), C.c_void_p(ptr.ctypes.data),
                          C.c_void_p(idx.ctypes.data), C.c_void_p(val.ctypes.data), layout, base, ptr.dtype.itemsize))


class BlockSymbolicSetup:
    def __init__(self, solver):
        self.solver = solver


class BlockNumericalSetup:
    """BlockDiagonalSolverNS / BlockTriangularSolverNS (BlockTriangularSolvers.jl:132-152) on a native handle.
    `mat` is the block system matrix as a nested list (None = zero block)."""

    _HALF = {"diagonal": abi.BLOCK_DIAGONAL, "lower": abi.BLOCK_LOWER, "upper": abi.BLOCK_UPPER}

    def __init__(self, solver, mat, device_id=None):
        lib = abi.load()
        self.solver, self._lib = solver, lib
        nb = len(solver.solvers)
        if len(mat) != nb or any(len(r) != nb for r in mat):
            raise ValueError("block matrix does not match the number of solvers")
        sizes = np.zeros(nb, dtype=np.int64)
        for i in range(nb):
            row = [m for m in mat[i] if m is not None]
            if not row:
                raise ValueError(f"block row {i} is empty")
            sizes[i] = int(_csr_fields(row[0])[0][0])
        self.sizes, self.n = sizes, int(sizes.sum())
        if device_id is None:
            device_id = 0
            try:
                import torch
                if torch.cuda.is_available():
                    device_id = torch.cuda.current_device()
            except Exception:
                pass
        h = C.c_void_p()
        abi.check_block(None, lib.gmg_block_create(C.byref(h), nb, C.c_void_p(sizes.ctypes.data), self._HALF[solver.half], device_id))
        self.h = h
        self.block_ns = [None] * nb
        try:
            for i in range(nb):
                for j in range(nb):
                    if mat[i][j] is not None:
                        _set_block(lib.gmg_block_set_system_block, h, i, j, mat[i][j])
                    blk = solver.blocks[i][j]
                    if i != j:
                        if isinstance(blk, MatrixBlock):
                            _set_block(lib.gmg_block_set_precond_block, h, i, j, blk.mat)
                        abi.check_block(h, lib.gmg_block_set_coeff(h, i, j, float(solver.coeffs[i][j])))
            for i, sv in enumerate(solver.solvers):
                blk = solver.blocks[i][i]
                Mi = blk.mat if isinstance(blk, MatrixBlock) else mat[i][i]
                if isinstance(sv, GMGLinearSolver):
                    g = GMGNumericalSetup(sv, Mi, device_id)
                    self.block_ns[i] = g
                    abi.check_block(h, lib.gmg_block_set_diag_gmg(h, i, g.h))
                    continue
                if isinstance(sv, CGSolver) and isinstance(sv.Pl, JacobiLinearSolver) and not sv.flexible:
                    kind, (mi, at, rt) = abi.BLOCK_CG_JACOBI, (sv.log.maxiter, sv.log.atol, sv.log.rtol)
                elif isinstance(sv, LUSolver):
                    kind, (mi, at, rt) = abi.BLOCK_LU, (0, 0.0, 0.0)
                elif isinstance(sv, JacobiLinearSolver):
                    kind, (mi, at, rt) = abi.BLOCK_JACOBI, (0, 0.0, 0.0)
                else:
                    raise NotImplementedError("block solvers on the device: GMGLinearSolver, CGSolver(JacobiLinearSolver()), "
                                              "LUSolver(), JacobiLinearSolver()")
                abi.check_block(h, lib.gmg_block_set_diag_solver(h, i, kind, mi, at, rt))
                if isinstance(blk, MatrixBlock):
                    shape, ptr, idx, val, layout, base = _csr_fields(blk.mat)
                    if ptr.dtype != idx.dtype:
                        ptr = ptr.astype(np.int64); idx = idx.astype(np.int64)
                    abi.check_block(h, lib.gmg_block_set_diag_matrix(h, i, shape[0], int(val.size), C.c_void_p(ptr.ctypes.data),
                                                                     C.c_void_p(idx.ctypes.data), C.c_void_p(val.ctypes.data),
                                                                     layout, base, ptr.dtype.itemsize))
            abi.check_block(h, lib.gmg_block_setup(h))
        except Exception:
            self.close()
            raise

    def _fill_logs(self):
        """copy the block solvers' ConvergenceLogs back (their last application)"""
        for i, sv in enumerate(self.solver.solvers):
            if isinstance(sv, (GMGLinearSolver, CGSolver)):
                res = abi.Result()
                abi.check_block(self.h, self._lib.gmg_block_diag_log(self.h, i, C.byref(res)))
                sv.log.num_iters, sv.log.flag = int(res.niters), int(res.flag)

    def mul(self, y, x):
        """mul!(y, A, x) on the block system"""
        px, ms, _k1 = _vec(x, self.n)
        py, ms2, _k2 = _vec(y, self.n, writable=True)
        if ms != ms2:
            raise TypeError("x and y must live in the same memory space")
        abi.check_block(self.h, self._lib.gmg_block_apply_system(self.h, px, py, ms))
        return y

    def close(self):
        if getattr(self, "h", None):
            self._lib.gmg_block_destroy(self.h)       # before the GMG handles it borrows
            self.h = None
        for g in getattr(self, "block_ns", []):
            if g is not None:
                g.close()
        self.block_ns = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _KrylovSymbolicSetup:
    def __init__(self, solver):
        self.solver = solver


class _KrylovNumericalSetup:
    """CGNumericalSetup / FGMRESNumericalSetup: holds the preconditioner's numerical
    setup (CGSolvers.jl:50-55, FGMRESSolvers.jl:96-102)."""

    def __init__(self, solver, A, device_id=None):
        self.solver = solver
        P = solver.Pr if isinstance(solver, FGMRESSolver) else solver.Pl
        if isinstance(P, (BlockDiagonalSolver, BlockTriangularSolver)):
            self.pc_kind = 1
            self.P_ns = BlockNumericalSetup(P, A, device_id)
            self.n = self.P_ns.n
            return
        if isinstance(P, GMGLinearSolver):
            self.pc_kind, gmg = 1, P
        elif isinstance(P, tuple) and len(P) == 2 and isinstance(P[1], GMGLinearSolver) and \
                (P[0] is None or isinstance(P[0], (JacobiLinearSolver, LinearSolverFromSmoother))):
            # (None | JacobiLinearSolver() | LinearSolverFromSmoother(gmg.pre_smoothers[0]), gmg):
            # Krylov on the handle's finest matrix with Pl = nothing / Jacobi / the finest pre-smoother
            if isinstance(P[0], LinearSolverFromSmoother):
                if P[0].smoother is not P[1].pre_smoothers[0]:
                    raise ValueError("LinearSolverFromSmoother must wrap the GMG's finest pre-smoother")
                self.pc_kind, gmg = 3, P[1]
            else:
                self.pc_kind, gmg = (0 if P[0] is None else 2), P[1]
        else:
            raise NotImplementedError("the device Krylov solvers take a GMGLinearSolver preconditioner "
                                      "(or (None|JacobiLinearSolver(), gmg) to reuse its finest matrix)")
        self.P_ns = GMGNumericalSetup(gmg, A, device_id)   # numerical_setup(symbolic_setup(Pl,A),A)
        self.n = self.P_ns.n


def symbolic_setup(solver, A=None):
    """Gridap.Algebra.symbolic_setup(solver, A)."""
    if isinstance(solver, GMGLinearSolver):
        return GMGSymbolicSetup(solver)
    if isinstance(solver, (CGSolver, FGMRESSolver, RichardsonLinearSolver)):
        return _KrylovSymbolicSetup(solver)
    if isinstance(solver, (BlockDiagonalSolver, BlockTriangularSolver)):
        return BlockSymbolicSetup(solver)
    raise TypeError(f"no symbolic_setup for {type(solver).__name__}")


def numerical_setup(ss, A=None, device_id=None):
    """Gridap.Algebra.numerical_setup(ss, A)."""
    if isinstance(ss, GMGSymbolicSetup):
        return GMGNumericalSetup(ss.solver, A, device_id)
    if isinstance(ss, _KrylovSymbolicSetup):
        return _KrylovNumericalSetup(ss.solver, A, device_id)
    if isinstance(ss, BlockSymbolicSetup):
        return BlockNumericalSetup(ss.solver, A, device_id)
    raise TypeError(f"no numerical_setup for {type(ss).__name__}")


def numerical_setup_(ns, A, smatrices=None):
    """Gridap.Algebra.numerical_setup!(ns, A) (smatrices: re-assembled matrices of all levels, optional)."""
    if isinstance(ns, GMGNumericalSetup):
        return ns.update(A, smatrices)
    if isinstance(ns, _KrylovNumericalSetup):
        ns.P_ns.update(A, smatrices)
        return ns
    raise TypeError(f"no numerical_setup! for {type(ns).__name__}")


def solve_(x, ns, b):
    """Gridap.Algebra.solve!(x, ns, b): in place on x, returns x."""
    if isinstance(ns, GMGNumericalSetup):
        log = ns.solver.log
        pb, ms, _kb = _vec(b, ns.n)
        px, ms2, _kx = _vec(x, ns.n, writable=True)
        if ms != ms2:
            raise TypeError("x and b must live in the same memory space")
        if ms == abi.MEM_HOST and ns.solver.pin_vectors:
            ns.pin(*(k for k, u in ((_kb, b), (_kx, x)) if k is u))       # the caller's own arrays only, never a conversion copy
        res = abi.Result()
        hist = np.zeros(log.maxiter + 1)
        abi.check(ns.h, ns._lib.gmg_apply(ns.h, pb, px, ms, C.byref(res), C.c_void_p(hist.ctypes.data), hist.size))
        log._fill(res, hist)
        return x
    if isinstance(ns, BlockNumericalSetup):
        pb, ms, _kb = _vec(b, ns.n)
        px, ms2, _kx = _vec(x, ns.n, writable=True)
        if ms != ms2:
            raise TypeError("x and b must live in the same memory space")
        abi.check_block(ns.h, ns._lib.gmg_block_precond_apply(ns.h, pb, px, ms))
        ns._fill_logs()
        return x
    if isinstance(ns, _KrylovNumericalSetup) and isinstance(ns.P_ns, BlockNumericalSetup):
        s, g = ns.solver, ns.P_ns
        log = s.log
        pb, ms, _kb = _vec(b, ns.n)
        px, ms2, _kx = _vec(x, ns.n, writable=True)
        if ms != ms2:
            raise TypeError("x and b must live in the same memory space")
        res = abi.Result()
        hist = np.zeros(log.maxiter + 1)
        if isinstance(s, CGSolver):
            abi.check_block(g.h, g._lib.gmg_block_cg_solve(g.h, pb, px, ms, log.maxiter, log.atol, log.rtol, int(s.flexible),
                                                           ns.pc_kind, C.byref(res), C.c_void_p(hist.ctypes.data), hist.size))
        else:
            abi.check_block(g.h, g._lib.gmg_block_fgmres_solve(g.h, pb, px, ms, s.m, int(s.restart), s.m_add, log.maxiter,
                                                               log.atol, log.rtol, ns.pc_kind, C.byref(res),
                                                               C.c_void_p(hist.ctypes.data), hist.size))
        log._fill(res, hist)
        g._fill_logs()
        return x
    if isinstance(ns, _KrylovNumericalSetup):
        s, g = ns.solver, ns.P_ns
        log = s.log
        pb, ms, _kb = _vec(b, ns.n)
        px, ms2, _kx = _vec(x, ns.n, writable=True)
        if ms != ms2:
            raise TypeError("x and b must live in the same memory space")
        if ms == abi.MEM_HOST and g.solver.pin_vectors:
            g.pin(*(k for k, u in ((_kb, b), (_kx, x)) if k is u))
        res = abi.Result()
        hist = np.zeros(log.maxiter + 1)
        if isinstance(s, CGSolver):
            abi.check(g.h, g._lib.gmg_cg_solve(g.h, pb, px, ms, log.maxiter, log.atol, log.rtol, int(s.flexible), ns.pc_kind,
                                               C.byref(res), C.c_void_p(hist.ctypes.data), hist.size))
        elif isinstance(s, RichardsonLinearSolver):
            abi.check(g.h, g._lib.gmg_richardson_solve(g.h, pb, px, ms, s.omega, log.maxiter, log.atol, log.rtol, ns.pc_kind,
                                                       C.byref(res), C.c_void_p(hist.ctypes.data), hist.size))
        else:
            pl = getattr(s, "Pl", None)
            pl_kind = 0 if pl is None else (2 if isinstance(pl, JacobiLinearSolver) else 3)
            abi.check(g.h, g._lib.gmg_fgmres_solve_pl(g.h, pb, px, ms, s.m, int(s.restart), s.m_add, log.maxiter,
                                                      log.atol, log.rtol, ns.pc_kind, pl_kind, C.byref(res),
                                                      C.c_void_p(hist.ctypes.data), hist.size))
        log._fill(res, hist)
        if ns.pc_kind == 1:
            g.fill_log()
        return x
    raise TypeError(f"no solve! for {type(ns).__name__}")


def mul_(y, op, x):
    """LinearAlgebra.mul!(y, op, x) for op = (ns, lev, 'A'|'P'|'R')."""
    ns, lev, which = op
    return ns.op_apply(lev, {"A": abi.OP_A, "P": abi.OP_P, "R": abi.OP_R}[which], x, y)
