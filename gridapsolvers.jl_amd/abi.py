"""ctypes binding of include/gmg_amd.h (libgmgamd.so).  Thin: no logic here."""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

OK = 0
ERR_INVALID, ERR_HIP, ERR_STATE, ERR_ALLOC, ERR_COMM, ERR_UNSUPPORTED, ERR_SINGULAR = 1, 2, 3, 4, 5, 6, 7
CSR, CSC = 0, 1
MEM_HOST, MEM_DEVICE = 0, 1
MODE_PRECONDITIONER, MODE_SOLVER = 0, 1
V_CYCLE, W_CYCLE, F_CYCLE = 0, 1, 2
CONVERGED_ATOL, CONVERGED_RTOL, DIVERGED_MAXITER, DIVERGED_BREAKDOWN = 0, 1, 2, 3
PRE, POST, PRE_AND_POST = 0, 1, 2
PATCH_LU, PATCH_NOPIVOT = 0, 1
OP_A, OP_P, OP_R = 0, 1, 2
LEVEL_KRYLOV = -1          # GMG_LEVEL_KRYLOV: the separate Krylov operator of a finest level in the overlapping layout
COARSE_DENSE_INVERSE, COARSE_CG_JACOBI, COARSE_HOST_CALLBACK = 0, 1, 2
BLOCK_DIAGONAL, BLOCK_LOWER, BLOCK_UPPER = 0, 1, 2
BLOCK_GMG, BLOCK_CG_JACOBI, BLOCK_LU, BLOCK_JACOBI = 1, 2, 3, 4


class Result(C.Structure):
    _fields_ = [("niters", C.c_int32), ("flag", C.c_int32), ("res0", C.c_double), ("res", C.c_double)]


class RedistPlan(C.Structure):
    """gmg_redist_plan: one direction of the redistribution between the glued and the subset partition of a level"""
    _fields_ = [("nnbr", C.c_int), ("nbr_rank", C.c_void_p), ("snd_ptr", C.c_void_p), ("snd_idx", C.c_void_p), ("rcv_ptr", C.c_void_p),
                ("rcv_idx", C.c_void_p), ("nself", C.c_int64), ("self_src", C.c_void_p), ("self_dst", C.c_void_p)]

    @classmethod
    def from_dict(cls, d, keep):
        """d: partition.build_local_hierarchy(...)["sub"]["to_sub" | "from_sub"]; `keep` collects the arrays the pointers refer to"""
        import numpy as np
        nbr = np.ascontiguousarray(d["nbr_rank"], dtype=np.int32)
        a = [np.ascontiguousarray(d[k], dtype=np.int64) for k in ("snd_ptr", "snd_idx", "rcv_ptr", "rcv_idx", "self_src", "self_dst")]
        keep += [nbr] + a
        return cls(int(nbr.size), nbr.ctypes.data, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data,
                   int(a[4].size), a[4].ctypes.data, a[5].ctypes.data)


class KernelStats(C.Structure):
    _fields_ = [("launches", C.c_int64), ("total_ms", C.c_double), ("alg_bytes", C.c_double),
                ("rows", C.c_int64), ("nnz", C.c_int64), ("layout_bytes", C.c_double), ("fused_passes", C.c_int64)]


# every symbol include/gmg_amd.h declares (tests check the library exports all of them)
SYMBOLS = {
    "gmg_create": [C.POINTER(C.c_void_p), C.c_int, C.c_int],
    "gmg_destroy": [C.c_void_p],
    "gmg_last_error": [C.c_void_p],
    "gmg_version": [],
    "gmg_set_matrix": [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                       C.c_int, C.c_int, C.c_int],
    "gmg_set_operator_rows": [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_int, C.c_int],
    "gmg_set_operator_rows_repeat": [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64],
    "gmg_update_values": [C.c_void_p, C.c_int, C.c_void_p],
    "gmg_update_values_csc": [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int],
    "gmg_set_prolongation": [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_int, C.c_int, C.c_int],
    "gmg_set_restriction": [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                            C.c_void_p, C.c_int, C.c_int, C.c_int],
    "gmg_set_smoother_jacobi": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double],
    "gmg_set_smoother_patch": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int64, C.c_void_p,
                               C.c_void_p, C.c_int, C.c_int],
    "gmg_set_smoother_patch_matrices": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int64, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p],
    "gmg_set_prolongation_patch_correction": [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_int],
    "gmg_set_prolongation_patch_correction_rhs": [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_int, C.c_int, C.c_int],
    "gmg_set_coarse_solver": [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p],
    "gmg_get_coarse_log": [C.c_void_p, C.POINTER(Result)],
    "gmg_set_options": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double],
    "gmg_set_option": [C.c_void_p, C.c_char_p, C.c_double],
    "gmg_get_option": [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int)],
    "gmg_host_register": [C.c_void_p, C.c_void_p, C.c_int64],
    "gmg_host_unregister": [C.c_void_p, C.c_void_p],
    "gmg_get_host_io_stats": [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
    "gmg_get_persist_retries": [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int)],
    "gmg_setup": [C.c_void_p],
    "gmg_set_verbose": [C.c_void_p, C.c_int],
    "gmg_set_stream": [C.c_void_p, C.c_void_p],
    "gmg_get_stream": [C.c_void_p, C.POINTER(C.c_void_p)],
    "gmg_get_log": [C.c_void_p, C.POINTER(Result), C.c_void_p, C.c_int],
    "gmg_apply": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Result), C.c_void_p, C.c_int],
    "gmg_cg_solve": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int,
                     C.POINTER(Result), C.c_void_p, C.c_int],
    "gmg_fgmres_solve": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                         C.c_double, C.c_double, C.c_int, C.POINTER(Result), C.c_void_p, C.c_int],
    "gmg_fgmres_solve_pl": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                            C.c_double, C.c_double, C.c_int, C.c_int, C.POINTER(Result), C.c_void_p, C.c_int],
    "gmg_richardson_solve": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_double, C.c_double, C.c_int,
                             C.POINTER(Result), C.c_void_p, C.c_int],
    "gmg_op_apply": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int],
    "gmg_smooth": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int],
    "gmg_precond_apply": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int],
    "gmg_coarse_solve": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int],
    "gmg_dot": [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double)],
    "gmg_comm_unique_id": [C.c_char_p, C.c_char_p],
    "gmg_comm_init_rccl": [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int],
    "gmg_comm_selftest": [C.c_void_p, C.POINTER(C.c_double)],
    "gmg_comm_latency_probe": [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_double)],
    "gmg_comm_init_host": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
    "gmg_set_partition": [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                          C.c_void_p],
    "gmg_set_partition_overlap": [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p],
    "gmg_get_comm_stats": [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
    "gmg_get_comm_info": [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "gmg_set_replication": [C.c_void_p, C.c_int, C.c_void_p, C.c_int64],
    "gmg_set_krylov_map": [C.c_void_p, C.c_void_p, C.c_int64],
    "gmg_comm_set_loopback": [C.c_void_p, C.c_int],
    "gmg_set_partition_overlap_hints": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int],
    "gmg_set_redistribution": [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int64, C.POINTER(RedistPlan), C.POINTER(RedistPlan)],
    "gmg_profile_enable": [C.c_void_p, C.c_int, C.c_int],
    "gmg_get_kernel_stats": [C.c_void_p, C.POINTER(KernelStats)],
    "gmg_get_kernel_stats_by_variant": [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)],
    "gmg_model_bytes": [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "gmg_sweep_signature": [C.c_void_p, C.c_int, C.c_char_p, C.c_int],
    "gmg_level_format": [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                         C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "gmg_device_bytes": [C.c_void_p, C.POINTER(C.c_int64)],
    "gmg_stream_probe": [C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_double)],
    "gmg_stream_probe_read": [C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_double)],
    "gmg_block_create": [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int, C.c_int],
    "gmg_block_destroy": [C.c_void_p],
    "gmg_block_last_error": [C.c_void_p],
    "gmg_block_comm_init_rccl": [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int],
    "gmg_block_comm_init_host": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
    "gmg_block_comm_set_loopback": [C.c_void_p, C.c_int],
    "gmg_block_set_partition": [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "gmg_block_set_system_block": [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int, C.c_int, C.c_int],
    "gmg_block_set_precond_block": [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_int, C.c_int, C.c_int],
    "gmg_block_set_coeff": [C.c_void_p, C.c_int, C.c_int, C.c_double],
    "gmg_block_set_diag_gmg": [C.c_void_p, C.c_int, C.c_void_p],
    "gmg_block_set_diag_solver": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double],
    "gmg_block_set_diag_matrix": [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_int, C.c_int, C.c_int],
    "gmg_block_setup": [C.c_void_p],
    "gmg_block_precond_apply": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int],
    "gmg_block_apply_system": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int],
    "gmg_block_fgmres_solve": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_double, C.c_double, C.c_int, C.POINTER(Result), C.c_void_p, C.c_int],
    "gmg_block_cg_solve": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int,
                           C.POINTER(Result), C.c_void_p, C.c_int],
    "gmg_block_diag_log": [C.c_void_p, C.c_int, C.POINTER(Result)],
}

HOST_EXCHANGE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_double),
                               C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int64))
HOST_ALLREDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_int)
COARSE_SOLVE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double))

_LIB = None


class GmgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libgmgamd status {code}: {msg}")
        self.code = code


# gmg_set_stream: names for HIP's default streams (hipStream_t 0 cannot be passed: NULL means "the handle's own stream")
STREAM_LEGACY, STREAM_PER_THREAD = 1, 2


def stream_arg(stream):
    """None -> NULL (back to the handle's own stream); a torch.cuda.Stream / integer hipStream_t -> that stream; the null stream
    (cuda_stream == 0: torch.cuda.current_stream() / default_stream() unless the caller made another one current) -> hipStreamLegacy."""
    if stream is None:
        return C.c_void_p(None)
    ptr = int(getattr(stream, "cuda_stream", stream))
    return C.c_void_p(ptr if ptr != 0 else STREAM_LEGACY)


def load(path=None):
    """dlopen libgmgamd.so.  Fails loudly when the HIP extension is missing:
    there is no CPU fallback in the product."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = path or _build.lib_path()
    # PyTorch-ROCm bundles its own libamdhip64 / libhsa-runtime64.  Two HIP runtimes in one
    # process do not both see the GPU, and device pointers must be shared with torch tensors,
    # so when torch is importable let it load its runtime FIRST; libgmgamd's NEEDED
    # libamdhip64.so.7 then binds to the copy already in the process.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    if not os.path.exists(path):
        raise ImportError(f"{path} not built: run `python __graft_entry__.py build` (hipcc --offload-arch=gfx950)")
    lib = C.CDLL(path)
    for name, args in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_char_p if name in ("gmg_last_error", "gmg_block_last_error") else C.c_int
    _LIB = lib
    return lib


def check(handle, status):
    if status != OK:
        msg = load().gmg_last_error(handle)
        raise GmgError(status, msg.decode() if msg else "")
    return status


def check_block(handle, status):
    if status != OK:
        msg = load().gmg_block_last_error(handle)
        raise GmgError(status, msg.decode() if msg else "")
    return status
