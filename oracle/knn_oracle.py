"""
oracle/knn_oracle.py — Python face of the CPU oracle for the cosine top-k behind
``collection.query`` (reference call site
codd_dal/metrics/metrics_semantic_metadata_store.py:314-316, scoring :336).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of bench.py.  Nothing under ``codd_query_engine_amd/`` may
import it.

PARITY UNPINNED against ChromaDB (third-party, unpinned, absent — see the header of
knn_oracle.c and DESIGN.md §4).  Two restatements live here:

* ``search`` / ``search_keys`` — the C restatement (knn_oracle.c) with the canonical
  fp32 evaluation order; ids AND scores must match the GPU bit for bit.
* ``search_f64`` — a plain numpy fp64 brute force (no order games); the GPU's
  scores must sit within 2e-6 of it (fp32 storage) and its ranking may differ from
  the canonical one only between rows whose fp64 scores are closer than that.
"""

from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CODD_ORACLE_LIB_DIR: load the library from another build directory (oracle/Makefile `asan-test`: the sanitizer build)
_LIB_PATH = os.path.join(os.environ.get("CODD_ORACLE_LIB_DIR") or os.path.join(_HERE, "_build"), "libknn_oracle.so")

DT_F32, DT_BF16, DT_F16 = 0, 1, 2
_DT_BY_NAME = {"f32": DT_F32, "bf16": DT_BF16, "f16": DT_F16}

_lib = None


def build(force: bool = False) -> str:
    """Compile knn_oracle.c -> oracle/_build/libknn_oracle.so (gcc, seconds)."""
    src = os.path.join(_HERE, "knn_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s", "all"], check=True)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        c_p, i64, i32, u32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_uint32
        L.oracle_elems_per_chunk.argtypes = [i32]
        L.oracle_f32_to_bf16.argtypes = [c_p, c_p, i64]
        L.oracle_bf16_to_f32.argtypes = [c_p, c_p, i64]
        L.oracle_f32_to_f16.argtypes = [c_p, c_p, i64]
        L.oracle_f16_to_f32.argtypes = [c_p, c_p, i64]
        L.oracle_canon_dot.argtypes = [c_p, c_p, i32, i32]
        L.oracle_canon_dot.restype = ctypes.c_float
        L.oracle_normalize_rows.argtypes = [c_p, c_p, i64, i32, i32]
        L.oracle_search_keys.argtypes = [c_p, i32, i64, i32, c_p, i32, i32, u32, c_p]
        L.oracle_unpack_keys.argtypes = [c_p, i64, c_p, c_p]
        L.oracle_search.argtypes = [c_p, i32, i64, i32, c_p, i32, i32, c_p, c_p]
        L.oracle_merge_keys.argtypes = [c_p, i32, i32, i32, c_p]
        L.oracle_scores_f64.argtypes = [c_p, i32, i64, i32, c_p, c_p]
        L.oracle_search_fast_f32.argtypes = [c_p, i64, i32, c_p, i32, i32, c_p, c_p]
        L.oracle_num_threads.restype = i32
        _lib = L
    return _lib


def _ptr(a: np.ndarray) -> ctypes.c_void_p:
    assert a.flags["C_CONTIGUOUS"]
    return ctypes.c_void_p(a.ctypes.data)


def pad_dim(d: int) -> int:
    """Stored row length: the next multiple of 64 elements (zero padded)."""
    return (int(d) + 63) // 64 * 64


# ----------------------------------------------------------------------------- ingest

def normalize_rows(x: np.ndarray, dpad: int | None = None) -> np.ndarray:
    """fp32 [n,d] -> fp32 [n,dpad], rows scaled to unit L2 norm exactly as the ingest
    kernel does (canonical sum of squares, IEEE sqrt and divide); zero rows stay zero."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, d = x.shape
    dpad = pad_dim(d) if dpad is None else dpad
    out = np.empty((n, dpad), dtype=np.float32)
    lib().oracle_normalize_rows(_ptr(x), _ptr(out), n, d, dpad)
    return out


def to_storage(rows_f32: np.ndarray, dtype: str) -> np.ndarray:
    """Round normalised fp32 rows to the index's storage dtype (RNE).  bf16/f16 come
    back as uint16 bit patterns."""
    rows_f32 = np.ascontiguousarray(rows_f32, dtype=np.float32)
    if dtype == "f32":
        return rows_f32
    out = np.empty(rows_f32.shape, dtype=np.uint16)
    fn = lib().oracle_f32_to_bf16 if dtype == "bf16" else lib().oracle_f32_to_f16
    fn(_ptr(rows_f32), _ptr(out), rows_f32.size)
    return out


def widen(rows: np.ndarray, dtype: str) -> np.ndarray:
    """Storage rows -> fp32 values (exact)."""
    if dtype == "f32":
        return np.ascontiguousarray(rows, dtype=np.float32)
    rows = np.ascontiguousarray(rows, dtype=np.uint16)
    out = np.empty(rows.shape, dtype=np.float32)
    fn = lib().oracle_bf16_to_f32 if dtype == "bf16" else lib().oracle_f16_to_f32
    fn(_ptr(rows), _ptr(out), rows.size)
    return out


# ----------------------------------------------------------------------------- search

def canon_dot(q: np.ndarray, c: np.ndarray, dtype: str = "f32") -> float:
    q = np.ascontiguousarray(q, dtype=np.float32)
    c = np.ascontiguousarray(c, dtype=np.float32)
    E = lib().oracle_elems_per_chunk(_DT_BY_NAME[dtype])
    return float(lib().oracle_canon_dot(_ptr(q), _ptr(c), q.shape[-1], E))


def search_keys(rows: np.ndarray, dtype: str, queries: np.ndarray, k: int, row_base: int = 0) -> np.ndarray:
    """Canonical exhaustive top-k as packed u64 keys [B,k] (descending, 0 = empty)."""
    queries = np.ascontiguousarray(queries, dtype=np.float32)
    B, dpad = queries.shape
    n = rows.shape[0]
    assert rows.shape[1] == dpad and dpad % 64 == 0
    rows = np.ascontiguousarray(rows)
    keys = np.zeros((B, k), dtype=np.uint64)
    rc = lib().oracle_search_keys(_ptr(rows), _DT_BY_NAME[dtype], n, dpad, _ptr(queries), B, k, row_base, _ptr(keys))
    if rc != 0:
        raise ValueError("oracle_search_keys rejected its arguments")
    return keys


def unpack_keys(keys: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    dist = np.empty(keys.shape, dtype=np.float32)
    rows = np.empty(keys.shape, dtype=np.int64)
    lib().oracle_unpack_keys(_ptr(keys), keys.size, _ptr(dist), _ptr(rows))
    return dist, rows


def search(rows: np.ndarray, dtype: str, queries: np.ndarray, k: int) -> tuple[np.ndarray, np.ndarray]:
    """(distances fp32 [B,k] ascending, rows int64 [B,k], -1/inf padded)."""
    return unpack_keys(search_keys(rows, dtype, queries, k))


def merge_keys(keys: np.ndarray, k: int) -> np.ndarray:
    """Top-k of a [B,m] key multiset — the shard merge."""
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    B, m = keys.shape
    out = np.zeros((B, k), dtype=np.uint64)
    lib().oracle_merge_keys(_ptr(keys), B, m, k, _ptr(out))
    return out


def scores_f64(rows: np.ndarray, dtype: str, q: np.ndarray) -> np.ndarray:
    q = np.ascontiguousarray(q, dtype=np.float32)
    rows = np.ascontiguousarray(rows)
    out = np.empty(rows.shape[0], dtype=np.float64)
    lib().oracle_scores_f64(_ptr(rows), _DT_BY_NAME[dtype], rows.shape[0], rows.shape[1], _ptr(q), _ptr(out))
    return out


def search_f64(rows_f32: np.ndarray, queries: np.ndarray, k: int) -> tuple[np.ndarray, np.ndarray]:
    """numpy fp64 brute force over (widened) stored rows: (scores [B,k] desc, rows [B,k]);
    ties -> lower row.  Independent of the C code path on purpose."""
    R = np.asarray(rows_f32, dtype=np.float64)
    Q = np.asarray(queries, dtype=np.float64)
    S = Q @ R.T
    n = R.shape[0]
    kk = min(k, n)
    order = np.lexsort((np.broadcast_to(np.arange(n), S.shape), -S), axis=1)[:, :kk]
    sc = np.take_along_axis(S, order, axis=1)
    if kk < k:
        order = np.pad(order, ((0, 0), (0, k - kk)), constant_values=-1)
        sc = np.pad(sc, ((0, 0), (0, k - kk)), constant_values=-np.inf)
    return sc, order.astype(np.int64)


def search_fast_f32(rows_f32: np.ndarray, queries: np.ndarray, k: int) -> tuple[np.ndarray, np.ndarray]:
    """bench.py cpu_baseline: all-core fp32 scan (free summation order)."""
    rows_f32 = np.ascontiguousarray(rows_f32, dtype=np.float32)
    queries = np.ascontiguousarray(queries, dtype=np.float32)
    B, dpad = queries.shape
    dist = np.empty((B, k), dtype=np.float32)
    out_rows = np.empty((B, k), dtype=np.int64)
    lib().oracle_search_fast_f32(_ptr(rows_f32), rows_f32.shape[0], dpad, _ptr(queries), B, k, _ptr(dist), _ptr(out_rows))
    return dist, out_rows


def num_threads() -> int:
    return int(lib().oracle_num_threads())
