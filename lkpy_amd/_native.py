"""
ctypes binding of the C ABI declared in ``include/lkamd.h`` (``lkpy_amd/_lkamd.so``).

The product path has NO CPU fallback: if the HIP library is missing or there is no
GPU, the calls raise (:class:`BackendUnavailable`).  Loading the library and listing
its symbols works without a GPU (used by the CPU-only tests).
"""

from __future__ import annotations

import ctypes
import os
import re
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p
from pathlib import Path

_PKG = Path(__file__).resolve().parent
# LK_AMD_LIBRARY: load another build of the library (kernel-variant experiments)
LIB_PATH = Path(os.environ.get("LK_AMD_LIBRARY", _PKG / "_lkamd.so"))
HEADER_PATH = _PKG.parent / "include" / "lkamd.h"

LK_OK = 0
LK_E_INVALID = -1
LK_E_HIP = -2
LK_E_NOT_SPD = -3
LK_E_NAN_SIM = -4
LK_E_NOMEM = -5
LK_E_CANCELLED = -6

SOLVER_CHOLESKY = 0
SOLVER_CG = 1
SOLVER_AUTO = 2


class BackendUnavailable(RuntimeError):
    "The HIP extension or the GPU is missing; there is deliberately no CPU fallback."


_lib = None


def _declare(lib):
    vp = c_void_p
    sigs = {
        "lk_last_error": (c_char_p, []),
        "lk_version": (c_char_p, []),
        "lk_device_count": (c_int, []),
        "lk_padded_dim": (c_int32, [c_int32]),
        "lk_pad_rows": (c_int, [vp, c_int64, c_int32, c_int32, vp, c_int32, vp]),
        "lk_unpad_rows": (c_int, [vp, c_int64, c_int32, c_int32, vp, c_int32, vp]),
        "lk_task_ctl_create": (c_int, [POINTER(vp)]),
        "lk_task_ctl_destroy": (None, [vp]),
        "lk_task_ctl_cancel": (None, [vp]),
        "lk_task_ctl_cancelled": (c_int, [vp]),
        "lk_task_ctl_reset": (None, [vp]),
        "lk_task_ctl_progress": (c_int, [vp, POINTER(c_int64), POINTER(c_int64)]),
        "lk_als_plan_set_ctl": (c_int, [vp, vp]),
        "lk_iknn_plan_set_ctl": (c_int, [vp, vp]),
        "lk_iknn_plan_enable_timing": (c_int, [vp, c_int]),
        "lk_iknn_plan_get_timing": (c_int, [vp, POINTER(ctypes.c_double), POINTER(c_int32)]),
        "lk_argtopn_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int32]),
        "lk_gramian_workspace_bytes": (c_size_t, [c_int32]),
        "lk_gramian": (c_int, [vp, c_int64, c_int32, c_int32, c_float, vp, c_int32, vp, vp]),
        "lk_als_plan_create": (c_int, [POINTER(vp), vp, c_int, c_int64, c_int32, c_int32]),
        "lk_als_plan_create_ex": (c_int, [POINTER(vp), vp, c_int, c_int64, c_int32, c_int32,
                                          c_int32]),
        "lk_als_plan_destroy": (None, [vp]),
        "lk_als_plan_workspace_bytes": (c_size_t, [vp]),
        "lk_als_plan_solver": (c_int32, [vp]),
        "lk_als_plan_set_cg": (c_int, [vp, c_float, c_int32]),
        "lk_als_plan_cg_stats": (c_int, [vp, vp, vp, POINTER(c_int64), POINTER(c_int64)]),
        "lk_als_plan_short_rows": (c_int64, [vp]),
        "lk_als_plan_long_rows": (c_int64, [vp]),
        "lk_als_plan_yref": (vp, [vp, vp]),
        "lk_als_plan_woodbury_rows": (c_int64, [vp]),
        "lk_als_plan_set_z": (c_int, [vp, vp]),
        "lk_als_plan_set_z_workspace": (c_int, [vp, vp]),
        "lk_als_plan_set_rhs_workspace": (c_int, [vp, vp]),
        "lk_als_plan_set_z_shared": (c_int, [vp, vp, vp]),
        "lk_als_plan_set_z_leader": (c_int, [vp, c_int]),
        "lk_als_plan_z_flag": (vp, [vp, vp]),
        "lk_spd_inverse_workspace_bytes": (c_size_t, [c_int32]),
        "lk_spd_inverse": (c_int, [vp, c_int32, c_int32, vp, vp, vp, vp]),
        "lk_als_implicit_half_epoch": (
            c_int,
            [vp, vp, vp, vp, c_int64, c_int64, c_int32, vp, c_int32, vp, c_int32, vp, c_int32, vp,
             vp, vp],
        ),
        "lk_als_explicit_half_epoch": (
            c_int,
            [vp, vp, vp, vp, c_int64, c_int64, c_int32, vp, c_int32, vp, c_int32, c_float, vp, vp,
             vp],
        ),
        "lk_als_check_status": (c_int, [vp, vp, vp]),
        "lk_als_plan_enable_timing": (c_int, [vp, c_int]),
        "lk_als_plan_get_timing": (
            c_int, [vp, POINTER(ctypes.c_double), POINTER(ctypes.c_double), POINTER(c_int32)]
        ),
        "lk_iknn_plan_create": (c_int, [POINTER(vp), vp, vp, c_int, c_int64, c_int64]),
        "lk_iknn_plan_create_rows": (
            c_int, [POINTER(vp), vp, vp, c_int, c_int64, c_int64, c_int64, c_int64]
        ),
        "lk_iknn_plan_destroy": (None, [vp]),
        "lk_iknn_plan_workspace_bytes": (c_size_t, [vp]),
        "lk_iknn_build_count": (
            c_int, [vp, vp, vp, vp, vp, vp, vp, c_float, c_int64, vp, vp, POINTER(c_int64), vp]
        ),
        "lk_iknn_build_fill": (
            c_int, [vp, vp, vp, vp, vp, vp, vp, c_float, c_int64, vp, vp, vp, vp, vp]
        ),
        "lk_iknn_truncate_workspace_bytes": (c_size_t, [c_int64, c_int64]),
        "lk_iknn_truncate_count": (
            c_int,
            [vp, vp, vp, vp, c_int, vp, c_int64, c_int64, c_int64, c_int64, vp, vp,
             POINTER(c_int64), vp],
        ),
        "lk_iknn_truncate_fill": (c_int, [vp, vp, vp, c_int64, c_int64, vp, vp, vp, vp, vp]),
        "lk_iknn_score_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int32]),
        "lk_iknn_recommend_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int32,
                                                         c_int32]),
        "lk_iknn_recommend": (c_int, [vp, vp, vp, c_int64, c_int64, vp, vp, vp, vp, c_int32,
                                      c_int32, c_int32, c_int, vp, c_int64, vp, vp, vp, vp]),
        "lk_knn_score_last_stats": (None, [POINTER(c_int64)]),
        "lk_iknn_recommend_last_packed": (c_int, []),
        "lk_iknn_score_batch": (
            c_int,
            [vp, vp, vp, c_int64, c_int64, vp, vp, vp, vp, vp, c_int32, c_int32, vp, vp, vp, vp],
        ),
        "lk_uknn_score_batch": (
            c_int,
            [vp, vp, vp, c_int64, c_int64, c_int64, vp, vp, vp, vp, vp, c_int32, c_int32, vp, vp,
             vp, vp],
        ),
        "lk_csr_rows_dot": (
            c_int, [vp, c_int, vp, vp, c_int64, vp, c_int64, c_int64, vp, c_int64, vp]
        ),
        "lk_download": (c_int, [vp, vp, c_size_t, c_int32, vp]),
        "lk_download_warmup": (c_int, []),
        "lk_download_i32_narrow": (c_int, [vp, vp, c_int64, vp, c_int32, vp]),
        "lk_ease_gram": (c_int, [vp, vp, vp, vp, c_int64, c_float, vp, c_int64, vp]),
        "lk_ease_score_batch": (
            c_int, [vp, vp, c_int64, vp, c_int64, c_int64, vp, c_int64, vp]
        ),
        "lk_score_topk_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int32]),
        "lk_score_topk": (
            c_int,
            [vp, c_int32, c_int64, vp, c_int32, c_int64, c_int32, c_int32, vp, vp, vp, vp, vp, vp],
        ),
        "lk_score_dense": (
            c_int, [vp, c_int32, c_int64, vp, c_int32, c_int64, c_int32, vp, c_int64, vp]
        ),
        "lk_argtopn": (c_int, [vp, c_int64, c_int64, c_int32, vp, vp, vp]),
        "lk_iknn_prep_center": (c_int, [vp, c_int, vp, vp, c_int64, vp, vp, vp, vp]),
        "lk_iknn_prep_scale": (c_int, [vp, c_int, vp, vp, vp, c_int64, vp, vp, vp]),
        "lk_csr_transpose_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int]),
        "lk_csr_transpose": (
            c_int, [vp, c_int, vp, c_int64, c_int64, c_int64, vp, vp, vp, vp, c_size_t, vp]
        ),
        "lk_synth_zipf_rows": (c_int, [vp, c_int64, c_int64, ctypes.c_uint64, vp, c_int64, vp, vp]),
        "lk_csr_relabel": (c_int, [vp, c_int, vp, vp, c_int64, vp, vp, vp, vp, vp, vp]),
        "lk_csr_gather_rows": (
            c_int, [vp, c_int, vp, vp, c_int64, vp, vp, vp, c_float, vp, vp, vp]
        ),
        "lk_als_implicit_half_epoch_host": (
            c_int,
            [vp, c_int, vp, vp, c_int64, c_int64, c_int32, vp, vp, vp, c_int32, vp],
        ),
        "lk_als_implicit_half_epoch_host_ctl": (
            c_int,
            [vp, c_int, vp, vp, c_int64, c_int64, c_int32, vp, vp, vp, c_int32, vp, vp],
        ),
    }  # fmt: skip
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return sigs


def load(build_if_missing: bool = False):
    "Load ``_lkamd.so``; raises :class:`BackendUnavailable` if it is not built."
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        if build_if_missing:
            from .csrc.build import build

            build()
        else:
            raise BackendUnavailable(
                f"{LIB_PATH} not found: build it with `python -m lkpy_amd.csrc.build` "
                "(or __graft_entry__.build()); lkpy_amd has no CPU fallback"
            )
    # ONE HIP runtime per process: the PyTorch-ROCm wheel bundles its own libamdhip64 /
    # libhsa-runtime64, and whichever copy is mapped first serves everybody (same sonames).  If
    # this library came first it would pull /opt/rocm's copy and torch's bundled HSA runtime would
    # then find no device ("No HIP GPUs are available": seen when __graft_entry__.build() loaded
    # the library before smoke() imported torch).  torch is the device-memory plumbing of this
    # package anyway, so it is imported here, before the dlopen.
    import torch  # noqa: F401

    try:
        lib = ctypes.CDLL(str(LIB_PATH))
    except OSError as e:  # pragma: no cover
        raise BackendUnavailable(f"cannot load {LIB_PATH}: {e}") from e
    _declare(lib)
    _lib = lib
    return lib


def declared_symbols() -> list[str]:
    "Every function ``include/lkamd.h`` declares (used by the export test)."
    text = HEADER_PATH.read_text()
    text = re.sub(r"#if 0.*?#endif /\* planned \*/", "", text, flags=re.S)
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lk_[a-z0-9_]+)\s*\(", text)))


def last_error() -> str:
    return load().lk_last_error().decode()


def check(rc: int, what: str = "lkamd call"):
    "Map a C-ABI return code to the exception the reference raises for it."
    if rc == LK_OK:
        return
    msg = last_error()
    if rc == LK_E_NOT_SPD:
        # src/accel/als/implicit.rs:79 -> RuntimeError("ALS solve error: ...")
        raise RuntimeError(msg)
    if rc == LK_E_NAN_SIM:
        raise ValueError("similarity is null")  # src/accel/knn/accum.rs:146-151
    if rc == LK_E_INVALID:
        raise ValueError(f"{what}: {msg}")
    if rc == LK_E_CANCELLED:
        raise KeyboardInterrupt(msg)
    raise RuntimeError(f"{what} failed ({rc}): {msg}")


def require_gpu():
    "Raise unless the library is loaded and a HIP device is visible."
    lib = load()
    if lib.lk_device_count() < 1:
        raise BackendUnavailable("no HIP device visible; lkpy_amd has no CPU fallback")
    return lib
