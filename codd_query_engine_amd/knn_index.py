"""
DeviceKnnIndex — thin Python owner of one `codd_knn_index` (include/codd_knn.h).

torch is plumbing here: it provides device buffers, the current HIP stream and (in
sharded.py) torch.distributed.  All arithmetic happens inside libcodd_knn.so.
"""

from __future__ import annotations

import ctypes

import numpy as np

from . import native


def _torch():
    import torch

    return torch


def require_gpu(device: str = "cuda:0"):
    torch = _torch()
    if not torch.cuda.is_available():
        raise native.NativeLibraryError(
            "no GPU visible to this process: the k-NN path has no CPU fallback "
            "(tests inject their own engine; products need an MI355X)"
        )
    return torch.device(device)


class DeviceKnnIndex:
    """Device-resident, L2-normalised row store with exact cosine top-k search.

    Engine protocol consumed by knn_client.Collection:
        count() / upsert(slots, vecs) / search(queries, k) -> (dist, rows) as numpy.
    """

    def __init__(self, dim: int, dtype: str = "f32", device: str = "cuda:0"):
        if dtype not in native.DTYPE_CODES:
            raise ValueError(f"dtype must be one of {sorted(native.DTYPE_CODES)}")
        self._lib = native.load()
        self.device = require_gpu(device)
        self.dim = int(dim)
        self.dtype = dtype
        self._dev_index = self.device.index if self.device.index is not None else _torch().cuda.current_device()
        h = ctypes.c_void_p()
        native.check(
            self._lib.codd_knn_create(ctypes.byref(h), self._dev_index, self.dim, native.DTYPE_CODES[dtype], native.METRIC_COSINE),
            "codd_knn_create",
        )
        self._h = h
        pd = ctypes.c_int()
        native.check(self._lib.codd_knn_dim(self._h, None, ctypes.byref(pd), None), "codd_knn_dim")
        self.padded_dim = pd.value

    # ------------------------------------------------------------------ lifecycle
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.codd_knn_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):  # best effort
        try:
            self.close()
        except Exception:
            pass

    def _stream(self) -> ctypes.c_void_p:
        return ctypes.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------ ingest
    def count(self) -> int:
        out = ctypes.c_int64()
        native.check(self._lib.codd_knn_count(self._h, ctypes.byref(out)), "codd_knn_count")
        return out.value

    def reserve(self, rows: int) -> None:
        native.check(self._lib.codd_knn_reserve(self._h, int(rows)), "codd_knn_reserve")

    def upsert(self, slots, vecs, normalize: bool = True) -> None:
        """Host vectors [n, dim] fp32 into the given row slots (append or overwrite)."""
        slots = np.ascontiguousarray(slots, dtype=np.int64)
        vecs = np.ascontiguousarray(vecs, dtype=np.float32)
        if vecs.ndim != 2 or vecs.shape[1] != self.dim or slots.shape != (vecs.shape[0],):
            raise ValueError(f"expected vecs [n,{self.dim}] and slots [n], got {vecs.shape} / {slots.shape}")
        native.check(
            self._lib.codd_knn_upsert_host(self._h, slots.ctypes.data, vecs.ctypes.data, vecs.shape[0], int(bool(normalize))),
            "codd_knn_upsert_host",
        )

    def upsert_device(self, first_slot: int, vecs, normalize: bool = True) -> None:
        """Device tensor [n, dim] fp32 (contiguous) into slots [first_slot, first_slot+n)."""
        torch = _torch()
        if not (isinstance(vecs, torch.Tensor) and vecs.is_cuda and vecs.dtype == torch.float32 and vecs.is_contiguous()):
            raise ValueError("upsert_device wants a contiguous fp32 CUDA tensor")
        if vecs.dim() != 2 or vecs.shape[1] != self.dim:
            raise ValueError(f"expected [n,{self.dim}], got {tuple(vecs.shape)}")
        native.check(
            self._lib.codd_knn_upsert_device(self._h, int(first_slot), vecs.data_ptr(), vecs.shape[0], int(bool(normalize)), self._stream()),
            "codd_knn_upsert_device",
        )

    def read_rows(self, first: int = 0, n: int | None = None) -> np.ndarray:
        """Stored rows (normalised, padded) as fp32 or uint16 bit patterns."""
        n = self.count() - first if n is None else n
        out = np.empty((n, self.padded_dim), dtype=np.float32 if self.dtype == "f32" else np.uint16)
        native.check(self._lib.codd_knn_read_rows(self._h, int(first), int(n), out.ctypes.data), "codd_knn_read_rows")
        return out

    def load_rows(self, rows: np.ndarray, first_slot: int = 0) -> None:
        """Stored rows as read_rows() returned them (persisted index) into slots [first_slot, ...)."""
        want = np.float32 if self.dtype == "f32" else np.uint16
        rows = np.ascontiguousarray(rows, dtype=want)
        if rows.ndim != 2 or rows.shape[1] != self.padded_dim:
            raise ValueError(f"expected stored rows [n,{self.padded_dim}], got {rows.shape}")
        native.check(self._lib.codd_knn_load_rows(self._h, int(first_slot), rows.ctypes.data, rows.shape[0]), "codd_knn_load_rows")

    # ------------------------------------------------------------------ search
    def _queries_tensor(self, queries):
        torch = _torch()
        if isinstance(queries, torch.Tensor):
            q = queries.to(device=self.device, dtype=torch.float32).contiguous()
        else:
            q = torch.from_numpy(np.ascontiguousarray(queries, dtype=np.float32)).to(self.device)
        if q.dim() != 2 or q.shape[1] != self.dim:
            raise ValueError(f"expected queries [B,{self.dim}], got {tuple(q.shape)}")
        return q

    def search_tensors(self, queries, k: int):
        """Device in, device out: (dist fp32 [B,k] ascending, rows int64 [B,k], -1 padded)."""
        torch = _torch()
        q = self._queries_tensor(queries)
        B = q.shape[0]
        dist = torch.empty((B, k), dtype=torch.float32, device=self.device)
        rows = torch.empty((B, k), dtype=torch.int64, device=self.device)
        native.check(
            self._lib.codd_knn_search(self._h, q.data_ptr(), B, int(k), dist.data_ptr(), rows.data_ptr(), self._stream()),
            "codd_knn_search",
        )
        return dist, rows

    def search(self, queries, k: int):
        """numpy in, numpy out (the façade's path)."""
        dist, rows = self.search_tensors(queries, k)
        return dist.cpu().numpy(), rows.cpu().numpy()

    def search_keys(self, queries, k: int, row_base: int = 0):
        """Shard-local packed keys [B,k] (u64 bit patterns in an int64 tensor), descending."""
        torch = _torch()
        q = self._queries_tensor(queries)
        B = q.shape[0]
        keys = torch.empty((B, k), dtype=torch.int64, device=self.device)
        native.check(
            self._lib.codd_knn_search_keys(self._h, q.data_ptr(), B, int(k), int(row_base), keys.data_ptr(), self._stream()),
            "codd_knn_search_keys",
        )
        return keys

    def merge_keys(self, keys, k: int):
        """Top-k of [B,m] packed keys -> (keys [B,k], dist [B,k], rows [B,k]) on device."""
        return merge_keys(keys, k, self.device)

    def approx_scores(self, queries):
        """Raw approximate scores of the MFMA filter: [256, count] fp32 tensor (diagnostics)."""
        torch = _torch()
        q = self._queries_tensor(queries)
        out = torch.zeros((256, self.count()), dtype=torch.float32, device=self.device)
        native.check(
            self._lib.codd_knn_approx_scores(self._h, q.data_ptr(), q.shape[0], out.data_ptr(), self._stream()),
            "codd_knn_approx_scores",
        )
        return out

    # ------------------------------------------------------------------ knobs
    def set_option(self, key: str, value: int) -> None:
        native.check(self._lib.codd_knn_set_option(self._h, key.encode(), int(value)), f"set_option({key})")

    def stat(self, key: str) -> int:
        out = ctypes.c_int64()
        native.check(self._lib.codd_knn_get_stat(self._h, key.encode(), ctypes.byref(out)), f"get_stat({key})")
        return out.value


def merge_keys(keys, k: int, device=None):
    """codd_knn_merge_keys on a [B,m] int64 CUDA tensor of packed keys."""
    torch = _torch()
    lib = native.load()
    if not (isinstance(keys, torch.Tensor) and keys.is_cuda and keys.dtype == torch.int64 and keys.dim() == 2):
        raise ValueError("merge_keys wants a [B,m] int64 CUDA tensor")
    keys = keys.contiguous()
    dev = keys.device if device is None else torch.device(device)
    B, m = keys.shape
    out_keys = torch.empty((B, k), dtype=torch.int64, device=dev)
    dist = torch.empty((B, k), dtype=torch.float32, device=dev)
    rows = torch.empty((B, k), dtype=torch.int64, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    native.check(
        lib.codd_knn_merge_keys(dev.index or 0, keys.data_ptr(), B, m, int(k), out_keys.data_ptr(), dist.data_ptr(), rows.data_ptr(), stream),
        "codd_knn_merge_keys",
    )
    return out_keys, dist, rows


def merge_shards(gathered, world_size: int, k: int, device=None):
    """codd_knn_merge_shards on the [G*B, k_in] int64 CUDA tensor an all_gather of the ranks' [B, k_in] keys delivers."""
    torch = _torch()
    lib = native.load()
    if not (isinstance(gathered, torch.Tensor) and gathered.is_cuda and gathered.dtype == torch.int64 and gathered.dim() == 2
            and gathered.shape[0] % world_size == 0):
        raise ValueError("merge_shards wants a [G*B, k_in] int64 CUDA tensor")
    gathered = gathered.contiguous()
    dev = gathered.device if device is None else torch.device(device)
    B, k_in = gathered.shape[0] // world_size, gathered.shape[1]
    out_keys = torch.empty((B, k), dtype=torch.int64, device=dev)
    dist = torch.empty((B, k), dtype=torch.float32, device=dev)
    rows = torch.empty((B, k), dtype=torch.int64, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    native.check(
        lib.codd_knn_merge_shards(dev.index or 0, gathered.data_ptr(), int(world_size), B, k_in, int(k), out_keys.data_ptr(), dist.data_ptr(),
                                  rows.data_ptr(), stream),
        "codd_knn_merge_shards",
    )
    return out_keys, dist, rows
