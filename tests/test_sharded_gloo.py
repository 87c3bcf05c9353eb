"""world_size-2 (and 3, uneven shards) run of the row-sharded search over gloo on the CPU:
covers shard bounds, global-row keys, the all_gather layout and the merge.  The shard-local
search is the checker engine here (no GPU); on the GPU the same ShardedSearcher drives
DeviceKnnIndex over RCCL (bench.py --gpus N)."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from codd_query_engine_amd.sharded import ShardedSearcher, shard_bounds
from oracle import knn_oracle as o
from tests._oracle_engine import OracleEngine


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class TensorOracleEngine:
    """OracleEngine speaking tensors, as DeviceKnnIndex does."""

    def __init__(self, inner):
        self.inner = inner

    def search_keys(self, queries, k, row_base):
        keys = self.inner.search_keys(np.asarray(queries), k, row_base)
        return torch.from_numpy(keys.view(np.int64).copy())


def oracle_merge(keys_t, k):
    merged, d, r = OracleEngine.merge_keys(keys_t.numpy().view(np.uint64), k)
    return torch.from_numpy(merged.view(np.int64).copy()), torch.from_numpy(d), torch.from_numpy(r)


def make_data(n, d, B):
    rng = np.random.default_rng(77)
    raw = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((B, d)).astype(np.float32)
    if n > 10:
        raw[n // 2 + 3] = raw[5]  # a cross-shard exact tie: the lower GLOBAL row must win
        q[0] = raw[5]
    return raw, q


def worker(rank, world, port, n, d, B, k, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        raw, q = make_data(n, d, B)
        lo, hi = shard_bounds(n, world, rank)
        eng = OracleEngine(d)
        if hi > lo:
            eng.upsert(np.arange(hi - lo, dtype=np.int64), raw[lo:hi])
        searcher = ShardedSearcher(TensorOracleEngine(eng), row_base=lo, merge=oracle_merge)
        dd, rr = searcher.search(q, k)
        dd2, rr2 = searcher.search_async(q, k).result()  # host engine: runs synchronously, same collective order on all ranks
        assert torch.equal(dd, dd2) and torch.equal(rr, rr2)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), dist=dd.numpy(), rows=rr.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 1001), (3, 500), (2, 1)])
def test_sharded_search_equals_single_index(tmp_path, world, n):
    d, B, k = 64, 5, 10
    port = free_port()
    mp.spawn(worker, args=(world, port, n, d, B, k, str(tmp_path)), nprocs=world, join=True)
    raw, q = make_data(n, d, B)
    d_ref, r_ref = o.search(o.normalize_rows(raw), "f32", o.normalize_rows(q), k)
    for rank in range(world):
        got = np.load(tmp_path / f"rank{rank}.npz")
        assert np.array_equal(got["rows"], r_ref), f"rank {rank}"
        assert np.array_equal(got["dist"], d_ref), f"rank {rank}"
    if n > 10:
        assert r_ref[0, 0] == 5 and r_ref[0, 1] == n // 2 + 3  # tie across shards: lower global row first


def test_shard_bounds_cover_rows_exactly():
    for n in (0, 1, 7, 8, 9, 10_000_000):
        for g in (1, 2, 3, 4, 8):
            spans = [shard_bounds(n, g, r) for r in range(g)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(0 <= hi - lo <= -(-n // g) for lo, hi in spans)
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)
