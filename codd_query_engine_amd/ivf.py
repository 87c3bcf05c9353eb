"""
Coarse-IVF over a DeviceKnnIndex (BASELINE config 5: the small-batch, HBM-bound regime).

Build (offline, once per corpus):  spherical k-means on a sample of the stored rows, then every
row is assigned to its nearest centroid and the rows are regrouped by list.
  * the heavy part of every step — scores of 256 rows against all centroids — is the engine's own
    bf16 MFMA GEMM (`codd_knn_approx_scores` on an index that holds the centroids);
  * torch supplies argmax / index_add / sort on the device: bookkeeping around the GEMM, not search.
Search: `codd_knn_ivf_search` (HIP): exact top-nprobe lists per query, canonical exact scores over
those lists, original row slots; `nprobe == nlist` reproduces the flat search bit for bit.

The reference has no counterpart (ChromaDB answers with HNSW, store.py:63-68); results are
approximate by design and reported as recall@k against this engine's exact search.
"""

from __future__ import annotations

import ctypes
from typing import Optional

from . import native
from .knn_index import DeviceKnnIndex


def _assign(torch, centroid_index: DeviceKnnIndex, vecs) -> "torch.Tensor":
    """nearest centroid (by approximate cosine) of every row of `vecs` [n, dim] (device fp32)."""
    nlist = centroid_index.count()
    out = torch.empty(vecs.shape[0], dtype=torch.int64, device=vecs.device)
    for a in range(0, vecs.shape[0], 256):
        chunk = vecs[a : a + 256]
        scores = centroid_index.approx_scores(chunk)[: chunk.shape[0], :nlist]
        out[a : a + chunk.shape[0]] = torch.argmax(scores, dim=1)
    return out


def build_ivf(index: DeviceKnnIndex, nlist: int, iters: int = 8, sample: Optional[int] = None, seed: int = 7,
              chunk_rows: int = 262_144) -> dict:
    """Cluster the stored rows into `nlist` lists and install the layout on `index`.

    sample: rows used for training (default min(count, 64 * nlist)).  Returns list-size statistics.
    """
    import torch

    lib = native.load()
    n, dim, dev = index.count(), index.dim, index.device
    if n < nlist:
        raise ValueError(f"{n} rows cannot form {nlist} lists")
    stream = lambda: ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)  # noqa: E731

    def rows_f32(first: int, count: int):
        out = torch.empty((count, dim), dtype=torch.float32, device=dev)
        native.check(lib.codd_knn_copy_rows_f32(index._h, first, count, out.data_ptr(), stream()), "codd_knn_copy_rows_f32")
        return out

    g = torch.Generator(device="cpu").manual_seed(seed)
    m = min(n, sample or 64 * nlist)
    # training sample: evenly spaced blocks of rows (cheap to read back, representative of the whole corpus)
    nblk = max(1, min(m // 256, 4096))
    per = m // nblk
    starts = (torch.arange(nblk, dtype=torch.float64) * ((n - per) / max(nblk - 1, 1))).long().tolist()
    train = torch.cat([rows_f32(s, per) for s in starts])
    cent = train[torch.randperm(train.shape[0], generator=g)[:nlist].to(dev)].clone()

    for _ in range(iters):
        cidx = DeviceKnnIndex(dim, "f32", str(dev))
        cidx.upsert_device(0, cent.contiguous())
        a = _assign(torch, cidx, train)
        cidx.close()
        sums = torch.zeros((nlist, dim), dtype=torch.float32, device=dev).index_add_(0, a, train)
        counts = torch.bincount(a, minlength=nlist)
        empty = counts == 0
        if empty.any():  # reseed empty lists from random training rows
            sums[empty] = train[torch.randint(0, train.shape[0], (int(empty.sum()),), generator=g).to(dev)]
        cent = torch.nn.functional.normalize(sums, dim=1)

    # assign every stored row, chunk by chunk, then group rows by list (stable: ascending slot inside a list)
    cidx = DeviceKnnIndex(dim, "f32", str(dev))
    cidx.upsert_device(0, cent.contiguous())
    assign = torch.empty(n, dtype=torch.int64, device=dev)
    for a0 in range(0, n, chunk_rows):
        c = min(chunk_rows, n - a0)
        assign[a0 : a0 + c] = _assign(torch, cidx, rows_f32(a0, c))
    cidx.close()
    perm = torch.sort(assign, stable=True).indices.contiguous()
    sizes = torch.bincount(assign, minlength=nlist)
    offsets = torch.zeros(nlist + 1, dtype=torch.int64, device=dev)
    offsets[1:] = torch.cumsum(sizes, 0)
    native.check(
        lib.codd_knn_ivf_install(index._h, cent.contiguous().data_ptr(), nlist, perm.data_ptr(), offsets.data_ptr(), stream()),
        "codd_knn_ivf_install",
    )
    torch.cuda.synchronize(dev)
    return {"nlist": nlist, "rows": n, "min_list": int(sizes.min()), "max_list": int(sizes.max()), "mean_list": n / nlist,
            "train_rows": int(train.shape[0]), "iters": iters}


def search_ivf(index: DeviceKnnIndex, queries, k: int, nprobe: int, row_base: int = 0):
    """(dist [B,k] fp32 ascending, rows [B,k] int64 original slots, -1 padded) on the device."""
    import torch

    lib = native.load()
    q = index._queries_tensor(queries)
    B = q.shape[0]
    dist = torch.empty((B, k), dtype=torch.float32, device=index.device)
    rows = torch.empty((B, k), dtype=torch.int64, device=index.device)
    native.check(
        lib.codd_knn_ivf_search(index._h, q.data_ptr(), B, int(k), int(nprobe), int(row_base), None, dist.data_ptr(), rows.data_ptr(),
                                index._stream()),
        "codd_knn_ivf_search",
    )
    return dist, rows


def search_ivf_keys(index: DeviceKnnIndex, queries, k: int, nprobe: int, row_base: int = 0):
    """Shard-local half of a row-sharded IVF search: [B,k] packed keys carrying GLOBAL rows (as codd_knn_search_keys)."""
    import torch

    lib = native.load()
    q = index._queries_tensor(queries)
    B = q.shape[0]
    keys = torch.empty((B, k), dtype=torch.int64, device=index.device)
    native.check(
        lib.codd_knn_ivf_search(index._h, q.data_ptr(), B, int(k), int(nprobe), int(row_base), keys.data_ptr(), None, None, index._stream()),
        "codd_knn_ivf_search",
    )
    return keys


class IvfShardEngine:
    """What ShardedSearcher needs from an engine, answered by the shard's IVF lists (BASELINE configs[4]: every rank
    builds an IVF over its own rows; the exchange is the same all_gather of B*k keys as for the flat search)."""

    def __init__(self, index: DeviceKnnIndex, nprobe: int):
        self.index, self.nprobe = index, int(nprobe)

    def search_keys(self, queries, k: int, row_base: int = 0):
        return search_ivf_keys(self.index, queries, k, self.nprobe, row_base)
