"""small_batch_kernel (csrc/codd_knn.hip): ONE query answered in ONE launch — every wave streams its share of the int8 shadow
and keeps its best approximate scores, the last workgroup to finish re-scores the survivors exactly (BASELINE configs[1]:
1M x 768, B = 1).  An option ("small_batch_max" = 1), not the default: round 3 measured it slower than the chain it replaces.  Only SPEED may differ from the filter chain it replaces: ids and distances equal
the oracle's bit for bit, and a query whose candidates a workgroup may have dropped goes to the exact-scan fallback."""

import numpy as np
import pytest

from oracle import knn_oracle as o

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Index():
    import torch

    assert torch.cuda.is_available()
    from codd_query_engine_amd.knn_index import DeviceKnnIndex

    return DeviceKnnIndex


def build(Index, raw, dtype="f32"):
    ix = Index(raw.shape[1], dtype=dtype)
    ix.upsert(np.arange(raw.shape[0], dtype=np.int64), raw)
    for key in ("filter_min_rows", "filter_min_rows_small", "filter_min_batch"):
        ix.set_option(key, 1)
    ix.set_option("small_batch_max", 1)     # (off by default: measured slower than the chain)
    ix.set_option("shadow8_cooldown", 0)    # (a fallback must not send the next searches to the 2-byte filter: these tests count passes)
    return ix


def oracle_answer(raw, q, k, dtype):
    return o.search(o.to_storage(o.normalize_rows(raw), dtype), dtype, o.normalize_rows(q), k)


@pytest.mark.parametrize(
    "n,d,k,dtype",
    [
        (200_000, 768, 10, "f32"),    # configs[1]'s shape, scaled down
        (100_001, 768, 10, "f32"),    # ragged last block
        (60_000, 768, 64, "f32"),     # k = 64: every lane of the lists
        (50_000, 384, 1, "f32"),      # MiniLM width (3 K-steps), k = 1
        (70_000, 512, 10, "f32"),     # 4 K-steps
        (40_000, 1024, 10, "f16"),    # 8 K-steps, 2-byte rows (config 5's width)
        (40_000, 768, 10, "bf16"),
        (30_017, 700, 10, "f32"),     # rows of 700 elements: padded to 704 stored, 768 int8
        (20_000, 768, 10, "f32"),     # few rows per wave: most lists stay short
    ],
)
def test_small_batch_is_exact(Index, n, d, k, dtype):
    rng = np.random.default_rng(n + d)
    raw = rng.standard_normal((n, d)).astype(np.float32)
    ix = build(Index, raw, dtype)
    for it in range(3):
        q = rng.standard_normal((1, d)).astype(np.float32)
        if it == 0:
            q[0] = raw[n // 3] + 0.05 * rng.standard_normal(d).astype(np.float32)   # a query with a real neighbour
        dist, rows = ix.search(q, k)
        d_ref, i_ref = oracle_answer(raw, q, k, dtype)
        assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref), it
    assert ix.stat("small_batch_passes") == 3 and ix.stat("filter_passes") == 0
    assert ix.stat("fallback_queries") == 0, "random data must pass the margin test"
    ix.close()


def test_small_batch_equals_the_filter_chain_and_is_skipped_where_it_does_not_apply(Index):
    rng = np.random.default_rng(3)
    raw = rng.standard_normal((80_000, 768)).astype(np.float32)
    ix = build(Index, raw)
    q = rng.standard_normal((1, 768)).astype(np.float32)
    a = ix.search(q, 10)
    assert ix.stat("small_batch_passes") == 1
    ix.set_option("small_batch_max", 0)              # the six-launch chain
    b = ix.search(q, 10)
    assert ix.stat("small_batch_passes") == 1 and ix.stat("filter_passes") == 1
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    ix.set_option("small_batch_max", 1)
    ix.search(rng.standard_normal((2, 768)).astype(np.float32), 10)     # two queries: the chain
    ix.search(q, 100)                                                    # k > 64: the chain
    assert ix.stat("small_batch_passes") == 1
    ix.close()
    narrow = build(Index, rng.standard_normal((30_000, 256)).astype(np.float32))   # 2 K-steps: not instantiated
    narrow.search(rng.standard_normal((1, 256)).astype(np.float32), 10)
    assert narrow.stat("small_batch_passes") == 0
    narrow.close()


def test_a_dense_cluster_fails_the_margin_test_and_is_answered_by_the_fallback(Index):
    """6,000 rows closer to each other than the int8 slack, the query in their middle: more candidates than a wave publishes.
    The margin test must catch it (dropmax >= L' - eps) and hand the query to the exact scan."""
    rng = np.random.default_rng(17)
    n, d, k = 120_000, 768, 10
    raw = rng.standard_normal((n, d)).astype(np.float32)
    centre = rng.standard_normal(d).astype(np.float32)
    members = np.arange(20_000, 26_000)                                   # contiguous: a handful of workgroups see all of them
    raw[members] = centre + 2e-3 * rng.standard_normal((members.size, d)).astype(np.float32)
    ix = build(Index, raw)
    for q, fallbacks in ((centre[None, :], 1), (rng.standard_normal((1, d)).astype(np.float32), 1)):   # (an ordinary query afterwards stays on the fast path)
        dist, rows = ix.search(q, k)
        d_ref, i_ref = oracle_answer(raw, q, k, "f32")
        assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref)
        assert ix.stat("fallback_queries") == fallbacks
    assert ix.stat("small_batch_passes") == 2
    ix.close()


def test_zero_query_repeats_upserts_and_global_row_ids(Index):
    import torch

    rng = np.random.default_rng(23)
    n, d, k = 90_000, 768, 10
    raw = rng.standard_normal((n, d)).astype(np.float32)
    ix = build(Index, raw)
    for it in range(4):                                                   # the arrival ticket and the fallback queue are reused, never cleared by a launch
        q = rng.standard_normal((1, d)).astype(np.float32)
        if it == 1:
            q[0] = 0.0                                                    # an all-zero query: every score 0, ties by row
        if it == 2:
            q[0] = raw[it * 999 + 7]
        dist, rows = ix.search(q, k)
        d_ref, i_ref = oracle_answer(raw, q, k, "f32")
        assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref), it
        upd = rng.choice(n, size=300, replace=False)
        raw[upd] = rng.standard_normal((300, d)).astype(np.float32)
        ix.upsert(upd.astype(np.int64), raw[upd])
    assert ix.stat("small_batch_passes") == 4
    # the shard-local half of a sharded search: packed keys carrying global rows
    qt = torch.from_numpy(rng.standard_normal((1, d)).astype(np.float32)).cuda()
    keys = ix.search_keys(qt, k, row_base=1_000_000)
    _, dist, rows = ix.merge_keys(keys, k)
    d_l, r_l = ix.search_tensors(qt, k)
    assert torch.equal(rows, r_l + 1_000_000) and torch.equal(dist, d_l)
    ix.close()
