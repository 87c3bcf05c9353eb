"""
Row-sharded search across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no distributed code at all; this is the build's own
multi-GPU shape for the k-NN behind collection.query (store.py:314-316):

  * the corpus is cut into contiguous row blocks, rank g owns rows
    [g*ceil(N/G), min(N, (g+1)*ceil(N/G))) — no rank ever sees another rank's rows;
  * every rank receives the same B x d query block and runs the single-GPU search on its
    shard, emitting B x k packed keys that already carry GLOBAL row ids
    (key = ord(score) << 32 | ~global_row, codd_knn_search_keys);
  * ONE all_gather of B*k u64 per rank (80 B at B=1, 20 KB at B=256, k=10 — latency-bound,
    nowhere near the 153 GB/s of an xGMI link) gives every rank all partials;
  * an integer top-k of the G*k keys per query (codd_knn_merge_keys) finishes on every rank;
    ties resolve exactly as on one GPU because the order lives in the keys.
"""

from __future__ import annotations

from typing import Any, Callable, Optional


def shard_bounds(n_total: int, world_size: int, rank: int) -> tuple[int, int]:
    """[begin, end) of the rows rank `rank` owns."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad world_size / rank")
    per = -(-int(n_total) // world_size)
    begin = min(n_total, rank * per)
    return begin, min(n_total, begin + per)


class PendingSearch:
    """Handle of ShardedSearcher.search_async: the results live on a side stream until `.result()`."""

    def __init__(self, stream: Any, out: tuple):
        self._stream, self._out = stream, out

    def result(self) -> tuple:
        if self._stream is not None:
            import torch

            cur = torch.cuda.current_stream(self._out[0].device)
            cur.wait_stream(self._stream)
            for t in self._out:
                t.record_stream(cur)
            self._stream = None
        return self._out


class ShardedSearcher:
    """Glue between a shard-local engine and the process group.

    engine.search_keys(queries, k, row_base) -> int64 tensor [B,k] of packed keys on the
    engine's device; merge(keys [B,m], k) -> (keys, dist, rows).  Products pass a
    DeviceKnnIndex and leave `merge` unset (HIP merge kernel); the gloo tests pass the
    checker engine and its merge.
    """

    def __init__(self, engine: Any, row_base: int, group: Optional[Any] = None,
                 merge: Optional[Callable[[Any, int], tuple]] = None, always_gather: bool = False):
        import torch.distributed as dist

        self.engine = engine
        self.always_gather = always_gather  # exercise the collective even with one rank (rehearsals)
        self.row_base = int(row_base)
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._merge_gathered = None  # [G*B,k] as gathered -> results, without a transpose pass (HIP engines only)
        if merge is None:
            from .knn_index import merge_keys as hip_merge
            from .knn_index import merge_shards as hip_merge_shards

            merge = hip_merge
            self._merge_gathered = hip_merge_shards
        self._merge = merge
        self._streams: list = []
        self._turn = 0

    def search_keys_local(self, queries, k: int):
        return self.engine.search_keys(queries, k, self.row_base)

    def search_async(self, queries, k: int, depth: int = 2) -> "PendingSearch":
        """Issue one batch on a side stream and return at once; `.result()` makes the caller's stream wait for it.

        Consecutive batches go to `depth` alternating HIP streams, so the small kernels, the all_gather and the merge
        of batch i overlap the filter kernel of batch i+1 (the engine keeps one workspace per stream).  Every rank must
        issue its batches in the same order, as with `search`.  With a host engine (gloo tests) it runs synchronously.
        """
        import torch

        dev = getattr(queries, "device", None)
        if not (isinstance(queries, torch.Tensor) and dev is not None and dev.type == "cuda"):
            return PendingSearch(None, self.search(queries, k))
        if len(self._streams) != depth:
            self._streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
            self._turn = 0
        side = self._streams[self._turn]
        self._turn = (self._turn + 1) % depth
        side.wait_stream(torch.cuda.current_stream(dev))  # the queries are ready
        with torch.cuda.stream(side):
            out = self.search(queries, k)
        queries.record_stream(side)
        return PendingSearch(side, out)

    def search(self, queries, k: int):
        """(dist [B,k], global rows [B,k]) — identical on every rank."""
        import torch
        import torch.distributed as dist

        local = self.search_keys_local(queries, k)  # [B,k] int64
        if self.world_size == 1 and not self.always_gather:
            _, d, r = self._merge(local, k)
            return d, r
        B = local.shape[0]
        # rank-major concatenation along dim 0 (the layout both RCCL and gloo accept)
        gathered = torch.empty((self.world_size * B, k), dtype=torch.int64, device=local.device)
        dist.all_gather_into_tensor(gathered, local.contiguous(), group=self.group)
        if self._merge_gathered is not None:
            _, d, r = self._merge_gathered(gathered, self.world_size, k)
            return d, r
        merged_in = gathered.view(self.world_size, B, k).permute(1, 0, 2).reshape(B, self.world_size * k).contiguous()
        _, d, r = self._merge(merged_in, k)
        return d, r
