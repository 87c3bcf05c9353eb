"""
oracle/hnsw_cpu.py — ctypes face of oracle/hnsw_cpu.c: a from-scratch CPU HNSW with the reference's parameters
(space cosine, M = 16, construction_ef = 200, search_ef = 100;
codd_dal/metrics/metrics_semantic_metadata_store.py:63-68).

TEST / BASELINE INFRASTRUCTURE ONLY: imported by tests/ and by bench.py's cpu_baseline leg, never by the package.
It restates the published algorithm because chromadb / hnswlib are absent and cannot be installed; its speed is a
like-for-like ALGORITHM baseline (approximate graph search, reported with recall@10 against the exact answer), not a
measurement of hnswlib itself.
"""

from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CODD_ORACLE_LIB_DIR: load the library from another build directory (oracle/Makefile `asan-test`: the sanitizer build)
_LIB_PATH = os.path.join(os.environ.get("CODD_ORACLE_LIB_DIR") or os.path.join(_HERE, "_build"), "libhnsw_cpu.so")
REFERENCE_PARAMS = {"M": 16, "construction_ef": 200, "search_ef": 100}  # store.py:63-68

_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "hnsw_cpu.c")
        if not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
            subprocess.run(["make", "-s", "-C", _HERE, "all"], check=True)
        L = ctypes.CDLL(_LIB_PATH)
        c_p, i32 = ctypes.c_void_p, ctypes.c_int
        L.hnsw_build.argtypes = [c_p, i32, i32, i32, i32, ctypes.c_uint64, i32]
        L.hnsw_build.restype = c_p
        L.hnsw_search.argtypes = [c_p, c_p, i32, i32, i32, c_p, c_p, i32]
        L.hnsw_free.argtypes = [c_p]
        L.hnsw_max_level.argtypes = [c_p]
        L.hnsw_max_level.restype = i32
        _lib = L
    return _lib


class HnswIndex:
    """Cosine HNSW over L2-normalised fp32 rows (the rows array is kept alive by this object)."""

    def __init__(self, rows: np.ndarray, M: int = 16, construction_ef: int = 200, seed: int = 1, threads: int = 0):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        assert rows.ndim == 2 and rows.shape[0] >= 1
        self.rows = rows
        self._h = lib().hnsw_build(rows.ctypes.data, rows.shape[0], rows.shape[1], M, construction_ef, seed, threads)
        if not self._h:
            raise ValueError("hnsw_build rejected its arguments")

    def search(self, queries: np.ndarray, k: int, search_ef: int = 100, threads: int = 0):
        """(distances [nq, k] ascending (1 - cosine), ids [nq, k]) for L2-normalised queries."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        ids = np.empty((q.shape[0], k), dtype=np.int32)
        dist = np.empty((q.shape[0], k), dtype=np.float32)
        lib().hnsw_search(self._h, q.ctypes.data, q.shape[0], k, search_ef, ids.ctypes.data, dist.ctypes.data, threads)
        return dist, ids

    def max_level(self) -> int:
        return lib().hnsw_max_level(self._h)

    def close(self):
        if self._h:
            lib().hnsw_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def recall_at_k(ids: np.ndarray, exact_ids: np.ndarray) -> float:
    hit = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(ids, exact_ids))
    return hit / float(exact_ids.size)
