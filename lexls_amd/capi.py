"""ctypes loader for liblexls_hip.so (the C ABI declared in include/lexls_hip.h).

The library is built in-tree by ``lexls_amd.build.build_native()`` (hipcc, gfx950).  There is no
Python or CPU fallback for the compute path: if the shared object is missing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LEXLS_HIP_LIB", os.path.join(_HERE, "csrc", "liblexls_hip.so"))  # env override: kernel-tuning A/B builds

# every symbol include/lexls_hip.h declares (tests/test_capi_symbols.py checks the export table against the header)
SYMBOLS = [
    "lexls_last_error", "lexls_version", "lexls_device_count",
    "lexls_lse_create", "lexls_lse_destroy", "lexls_lse_set_stream", "lexls_lse_synchronize",
    "lexls_lse_set_tolerance", "lexls_lse_set_obj_dim", "lexls_lse_set_fixed", "lexls_lse_set_ctr_type",
    "lexls_lse_set_problem_host", "lexls_lse_set_problem_device", "lexls_lse_set_skip",
    "lexls_lse_set_constraint_data", "lexls_lse_gather_problem", "lexls_lse_solve_least_norm_2", "lexls_lse_set_fixed_type", "lexls_lse_get_fixed_type", "lexls_lse_set_deferred_sync", "lexls_lse_set_regularization", "lexls_lse_set_cg_iterations", "lexls_lsi_solve_ex", "lexls_lsi_solve_debug", "lexls_lsi_batch_solve_ex", "lexls_lse_solve_least_norm_3",
    "lexls_lsi_batch_create", "lexls_lsi_batch_run", "lexls_lsi_batch_destroy", "lexls_lsi_batch_stats",
    "lexls_lse_round_layout", "lexls_lse_upload_round", "lexls_lse_download_round", "lexls_lse_sensitivity_resident", "lexls_lse_set_sensitivity_scan",
    "lexls_lse_factorize", "lexls_lse_solve", "lexls_lse_factorize_solve", "lexls_lse_solve_least_norm",
    "lexls_lse_residual", "lexls_lse_sensitivity",
    "lexls_lse_get_x", "lexls_lse_get_factor", "lexls_lse_get_hh_scalars", "lexls_lse_get_permutation", "lexls_lse_get_ranks",
    "lexls_lse_get_v", "lexls_lse_get_mu", "lexls_lse_get_lambda", "lexls_lse_get_sensitivity", "lexls_lse_get_ctr_type",
    "lexls_lse_device_ptr", "lexls_lse_last_kernel", "lexls_lse_set_kernel_policy",
    "lexls_lse_set_prefix_reuse", "lexls_lse_prefix_reuse_ready", "lexls_lse_set_resume_levels",
    "lexls_lsi_solve", "lexls_lsi_solve_dat", "lexls_lsi_batch_solve",
]

ARRAY = dict(x=0, factor=1, hh=2, perm=3, rank=4, first_col=5, total_rank=6, v=7, lam=8, input=9)

_lib = None


class LexlsError(RuntimeError):
    pass


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LexlsError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        _lib.lexls_last_error.restype = C.c_char_p
        _lib.lexls_lse_last_kernel.restype = C.c_char_p
        _lib.lexls_lse_last_kernel.argtypes = [C.c_void_p]
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise LexlsError(f"liblexls_hip error {rc}: {lib().lexls_last_error().decode()}")
