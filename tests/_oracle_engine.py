"""Checker engine for HOST-LOGIC tests: satisfies the engine protocol of
codd_query_engine_amd.knn_client.Collection with the CPU oracle, so the façade, the store
and the sharded merge can be exercised without a GPU.  Lives under tests/ on purpose — the
product has no CPU search path."""

from __future__ import annotations

import numpy as np

from oracle import knn_oracle as o


class OracleEngine:
    def __init__(self, dim: int, dtype: str = "f32"):
        self.dim = int(dim)
        self.dtype = dtype
        self.padded_dim = o.pad_dim(dim)
        self._rows = np.zeros((0, self.padded_dim), dtype=np.float32 if dtype == "f32" else np.uint16)

    def count(self) -> int:
        return self._rows.shape[0]

    def upsert(self, slots, vecs, normalize: bool = True) -> None:
        slots = np.asarray(slots, dtype=np.int64)
        vecs = np.ascontiguousarray(vecs, dtype=np.float32)
        assert vecs.shape == (slots.shape[0], self.dim)
        need = int(slots.max()) + 1 if slots.size else 0
        if need > self._rows.shape[0]:
            grown = np.zeros((need, self.padded_dim), dtype=self._rows.dtype)
            grown[: self._rows.shape[0]] = self._rows
            self._rows = grown
        if normalize:
            normed = o.normalize_rows(vecs, self.padded_dim)
        else:
            normed = np.zeros((vecs.shape[0], self.padded_dim), dtype=np.float32)
            normed[:, : self.dim] = vecs
        self._rows[slots] = o.to_storage(normed, self.dtype)

    def read_rows(self, first: int = 0, n: int | None = None) -> np.ndarray:
        n = self.count() - first if n is None else n
        return self._rows[first : first + n].copy()

    def load_rows(self, rows, first_slot: int = 0) -> None:
        rows = np.ascontiguousarray(rows, dtype=self._rows.dtype)
        assert rows.shape[1] == self.padded_dim
        need = first_slot + rows.shape[0]
        if need > self._rows.shape[0]:
            grown = np.zeros((need, self.padded_dim), dtype=self._rows.dtype)
            grown[: self._rows.shape[0]] = self._rows
            self._rows = grown
        self._rows[first_slot:need] = rows

    def _prep(self, queries) -> np.ndarray:
        q = np.ascontiguousarray(queries, dtype=np.float32)
        assert q.ndim == 2 and q.shape[1] == self.dim
        return o.normalize_rows(q, self.padded_dim)

    def search(self, queries, k: int):
        return o.search(self._rows, self.dtype, self._prep(queries), k)

    def search_keys(self, queries, k: int, row_base: int = 0) -> np.ndarray:
        return o.search_keys(self._rows, self.dtype, self._prep(queries), k, row_base)

    @staticmethod
    def merge_keys(keys: np.ndarray, k: int):
        merged = o.merge_keys(keys, k)
        dist, rows = o.unpack_keys(merged)
        return merged, dist, rows
