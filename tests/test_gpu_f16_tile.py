"""The 2-byte (fp16) filter of full query blocks through the tile program of csrc/filter_i8.h (round 3): the same hand-ordered
schedule on v_mfma_f32_16x16x32_f16, taken when the int8 filter is off or cooling down (dense clusters) and the rows have 6, 12, ...
K-steps of 64 elements.  It may only change SPEED: ids and distances must equal the oracle's bit for bit, with and without it."""

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
    ix.set_option("shadow8", 0)   # the 2-byte filter
    return ix


def oracle_answer(raw, q, k, dtype):
    rows_ref = o.to_storage(o.normalize_rows(raw), dtype)
    return o.search(rows_ref, dtype, o.normalize_rows(q), k)


@pytest.mark.parametrize(
    "n,d,B,k,dtype",
    [
        (70_001, 768, 256, 10, "f32"),    # 12 K-steps, ragged last tile
        (70_001, 768, 129, 10, "f32"),    # the smallest batch that takes it
        (70_000, 768, 200, 100, "f32"),   # k = 100 (the filter needs 2k sample tiles)
        (50_000, 768, 256, 10, "bf16"),   # 2-byte stored rows (the shadow is fp16 whatever the storage type)
        (30_000, 384, 256, 10, "f32"),    # 6 K-steps: the static six-step form
        (40_000, 1152, 256, 10, "bf16"),  # 18 K-steps: run-time cursors
        (6_000, 768, 256, 10, "f32"),     # 24 tiles: most workgroups have none
        (140_000, 768, 256, 10, "f32"),   # two and three tiles per workgroup
    ],
)
def test_fp16_tile_filter_is_exact(Index, n, d, B, k, dtype):
    rng = np.random.default_rng(n + B + d)
    raw = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((B, d)).astype(np.float32)
    q[3] = raw[n - 1] + 0.05 * rng.standard_normal(d).astype(np.float32)   # a neighbour in the ragged last tile
    q[7] = 0.0
    q[8] = -raw[11]
    d_ref, i_ref = oracle_answer(raw, q, k, dtype)
    ix = build(Index, raw, dtype)
    for rep in range(2):
        dist, rows = ix.search(q, k)
        assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref), rep
    assert ix.stat("f16_tile_passes") == 2 and ix.stat("shadow8_passes") == 0
    hits_tile = ix.stat("filter_hits")
    ix.set_option("i8_pair", 1)           # the same rows through the tile program with run-time cursors
    dist, rows = ix.search(q, k)
    assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref)
    assert ix.stat("f16_tile_passes") == 3 and ix.stat("filter_hits") - hits_tile == hits_tile // 2
    ix.set_option("i8_pair", 2)
    hits_tile = 2 * (ix.stat("filter_hits") // 3)
    ix.set_option("f16_tile", 0)          # the first-generation kernel on the same shadow: same answer, same candidate lists
    dist, rows = ix.search(q, k)
    assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref)
    assert ix.stat("f16_tile_passes") == 3
    assert ix.stat("filter_hits") - 3 * (hits_tile // 2) == hits_tile // 2, "both kernels test approx >= thr[q] on the same scores"
    ix.close()


def test_fp16_tile_filter_on_a_dense_cluster(Index):
    """What the path exists for: a cluster tighter than the int8 slack that most of the batch points at.  With the int8 filter ON the
    engine measures the survivor volume, cools down to the 2-byte filter and takes the tile program there."""
    rng = np.random.default_rng(2718)
    n, d, B, k = 120_000, 768, 256, 10
    raw = rng.standard_normal((n, d)).astype(np.float32)
    centre = rng.standard_normal(d).astype(np.float32)
    centre /= np.linalg.norm(centre)
    members = rng.choice(n, size=60_000, replace=False)
    raw[members] = centre + 0.3 * rng.standard_normal((60_000, d)).astype(np.float32) / np.sqrt(d)
    q = rng.standard_normal((B, d)).astype(np.float32)
    q[:220] = centre + 0.3 * rng.standard_normal((220, d)).astype(np.float32) / np.sqrt(d)
    d_ref, i_ref = oracle_answer(raw, q, k, "f32")
    ix = build(Index, raw)
    ix.set_option("shadow8", 1)
    for rep in range(4):
        dist, rows = ix.search(q, k)
        assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref), rep
    assert ix.stat("f16_tile_passes") >= 1, "the cooldown should have moved the batch to the 2-byte filter"
    assert ix.stat("fallback_queries") == 0
    ix.close()
