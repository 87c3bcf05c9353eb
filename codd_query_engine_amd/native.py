"""
ctypes binding of include/codd_knn.h (the C ABI of the HIP library).

There is no CPU fallback: if libcodd_knn.so is missing or fails to load, or no GPU is
visible, every product entry point raises NativeLibraryError — loudly, on purpose.
"""

from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CODD_KNN_LIB selects an alternative build of the SAME source (kernel A/B experiments, scripts/)
LIB_PATH = os.environ.get("CODD_KNN_LIB") or os.path.join(_HERE, "csrc", "libcodd_knn.so")

DTYPE_CODES = {"f32": 0, "bf16": 1, "f16": 2}
DTYPE_NAMES = {v: k for k, v in DTYPE_CODES.items()}
METRIC_COSINE = 0
MAX_K = 128
MAX_BATCH = 1024

# every symbol include/codd_knn.h declares: (name, restype, argtypes)
_c_idx = ctypes.c_void_p
_i64p = ctypes.POINTER(ctypes.c_int64)
_intp = ctypes.POINTER(ctypes.c_int)
ABI = [
    ("codd_knn_version", ctypes.c_char_p, []),
    ("codd_knn_last_error", ctypes.c_char_p, []),
    ("codd_knn_create", ctypes.c_int, [ctypes.POINTER(_c_idx), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    ("codd_knn_destroy", ctypes.c_int, [_c_idx]),
    ("codd_knn_reserve", ctypes.c_int, [_c_idx, ctypes.c_int64]),
    ("codd_knn_upsert_host", ctypes.c_int, [_c_idx, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]),
    ("codd_knn_upsert_device", ctypes.c_int, [_c_idx, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    ("codd_knn_load_rows", ctypes.c_int, [_c_idx, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]),
    ("codd_knn_count", ctypes.c_int, [_c_idx, _i64p]),
    ("codd_knn_dim", ctypes.c_int, [_c_idx, _intp, _intp, _intp]),
    ("codd_knn_read_rows", ctypes.c_int, [_c_idx, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]),
    ("codd_knn_search", ctypes.c_int, [_c_idx, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    ("codd_knn_search_keys", ctypes.c_int, [_c_idx, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]),
    ("codd_knn_merge_keys", ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    ("codd_knn_merge_shards", ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    ("codd_knn_approx_scores", ctypes.c_int, [_c_idx, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    ("codd_knn_copy_rows_f32", ctypes.c_int, [_c_idx, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    ("codd_knn_ivf_install", ctypes.c_int, [_c_idx, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    ("codd_knn_ivf_search", ctypes.c_int, [_c_idx, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    ("codd_knn_set_option", ctypes.c_int, [_c_idx, ctypes.c_char_p, ctypes.c_int64]),
    ("codd_knn_get_stat", ctypes.c_int, [_c_idx, ctypes.c_char_p, _i64p]),
]


class NativeLibraryError(RuntimeError):
    """libcodd_knn.so is absent/unloadable, or a C-ABI call returned an error code."""


_lib = None


def load() -> ctypes.CDLL:
    """dlopen the in-tree HIP library and bind every ABI symbol.

    torch is imported first so that the process holds ONE HIP runtime: torch's wheel
    bundles libamdhip64 under the same SONAME the library was linked against, and the
    dynamic loader then resolves ours to the copy torch already mapped — device pointers
    and streams can cross between the two.
    """
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} not found: build it with `python -m codd_query_engine_amd.build` "
            "(there is no CPU fallback for the search path)"
        )
    try:
        import torch  # noqa: F401  (side effect: maps torch's HIP runtime first)
    except Exception:  # pragma: no cover - torch is part of the image
        pass
    try:
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    except OSError as e:
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, restype, argtypes in ABI:
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NativeLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def last_error() -> str:
    msg = load().codd_knn_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise NativeLibraryError(f"{what} failed with code {rc}: {last_error()}")
