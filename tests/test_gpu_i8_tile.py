"""i8_tile_kernel (csrc/filter_i8.h): the int8 filter GEMM of full query blocks (129..256 queries, and 65..128 through its
8-query-block instantiation; rows of more than 512
elements; from 384 elements on with the option i8v2 = 2).  Hand-ordered LDS-DMA staging, a lagging half of the
workgroup, deferred epilogues with an integer pre-test: all of it may only change SPEED.  Ids and distances must equal
the oracle's bit for bit, and the candidate lists must be the ones the first-generation kernel writes."""

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


def build(Index, raw, dtype="f32", i8v2=1):
    ix = Index(raw.shape[1], dtype=dtype)
    ix.upsert(np.arange(raw.shape[0], dtype=np.int64), raw)
    for key in ("filter_min_rows", "filter_min_rows_small", "filter_min_batch"):
        ix.set_option(key, 1)
    ix.set_option("shadow8", 1)
    ix.set_option("i8v2", i8v2)
    return ix


def oracle_answer(raw, q, k, dtype):
    rows_ref = o.to_storage(o.normalize_rows(raw), dtype)
    return o.search(rows_ref, dtype, o.normalize_rows(q), k)


@pytest.mark.parametrize(
    "n,d,B,k,dtype,i8v2",
    [
        (70_000, 768, 256, 10, "f32", 1),    # the headline shape, scaled down: 274 tiles, 6 K-steps
        (70_000, 768, 128, 10, "f32", 1),    # 65..128 queries: the 8-query-block instantiation (half the slice DMA per interval)
        (70_001, 768, 65, 10, "f32", 1),     # ... its smallest batch, ragged last tile
        (66_000, 640, 100, 100, "f32", 1),   # ... 5 K-steps (generic loop), k = 100
        (30_000, 1024, 128, 10, "bf16", 1),  # ... 8 K-steps
        (50_000, 384, 96, 10, "f32", 2),     # ... 3 K-steps, i8v2 = 2
        (70_001, 768, 129, 10, "f32", 1),    # ragged last tile, smallest batch that takes this kernel
        (40_000, 640, 200, 10, "f32", 1),    # 5 K-steps: tiles end at every position of the 3-interval body
        (40_000, 1024, 256, 10, "f16", 1),   # 8 K-steps (config 5's width)
        (20_000, 2048, 160, 10, "bf16", 1),  # widest row: 16 K-steps
        (66_000, 384, 256, 10, "f32", 2),    # 3 K-steps (the smallest this kernel takes), i8v2 = 2
        (66_000, 512, 255, 100, "f32", 2),   # 4 K-steps, k = 100
        (6_000, 768, 256, 10, "f32", 1),     # 24 tiles: most workgroups have no tile at all
        (140_000, 768, 256, 10, "f32", 1),   # 547 tiles: workgroups with 2 and with 3 tiles
    ],
)
def test_tile_kernel_is_exact(Index, n, d, B, k, dtype, i8v2):
    rng = np.random.default_rng(n + B + d)
    raw = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((B, d)).astype(np.float32)
    q[3] = raw[n // 2] + 0.05 * rng.standard_normal(d).astype(np.float32)  # a query with a real neighbour
    ix = build(Index, raw, dtype, i8v2)
    dist, rows = ix.search(q, k)
    assert ix.stat("i8v2_passes") == 1 and ix.stat("shadow8_passes") == 1
    d_ref, i_ref = oracle_answer(raw, q, k, dtype)
    assert np.array_equal(rows, i_ref)
    assert np.array_equal(dist, d_ref)
    assert ix.stat("fallback_queries") == 0, "random data must not need the fallback"
    ix.close()


@pytest.mark.parametrize(
    "n,d,B,pair",
    [
        (70_001, 768, 256, 2),    # the static six-step program (the default for 768 elements)
        (70_001, 768, 256, 1),    # the same rows through the pair program with run-time cursors
        (70_001, 768, 256, 0),    # ... and with a barrier behind every K-step
        (70_001, 768, 100, 2),    # 8 query blocks, static
        (70_001, 768, 100, 1),
        (40_000, 1536, 256, 2),   # 12 K-steps: the pair program's inner loop (no static form)
        (40_000, 1536, 128, 2),
        (5_000, 768, 256, 2),     # 20 tiles: most workgroups have none, the others exactly one (the static loop's first tile is its last)
        (131_072 + 300, 768, 200, 2),  # two and three tiles per workgroup, a ragged last one
    ],
)
def test_every_tile_program_is_exact(Index, n, d, B, pair):
    """"i8_pair" picks the tile program: 2 = the static six-step program where rows have exactly 6 K-steps (round 3: every cursor a
    compile-time constant, early corpus loads, the flush at the end of the tile), 1 = the pair program with run-time cursors, 0 = one
    barrier per K-step.  All of them must return the oracle's bits, whichever the default happens to be."""
    rng = np.random.default_rng(n + B + d + pair)
    raw = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((B, d)).astype(np.float32)
    q[5] = raw[n - 1] + 0.05 * rng.standard_normal(d).astype(np.float32)   # a neighbour in the ragged last tile
    q[6] = raw[0]
    dtype = "f32" if d <= 1024 else "bf16"   # (f32 rows stop at 1,024 elements)
    ix = build(Index, raw, dtype, 1)
    ix.set_option("i8_pair", pair)
    k = 10
    for rep in range(2):
        dist, rows = ix.search(q, k)
        d_ref, i_ref = oracle_answer(raw, q, k, dtype)
        assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref), rep
    assert ix.stat("i8v2_passes") == 2 and ix.stat("fallback_queries") == 0
    ix.close()


def test_static_program_flushes_a_full_hit_list_at_the_end_of_a_tile(Index):
    """A cluster spread over many tiles that 200 queries point at: the workgroup's LDS hit list passes its flush mark in the
    middle of the launch, so the static program's flush (behind the tile's last barrier, in front of its epilogue, with its own
    closing barrier) runs many times; the candidates it hands over must be complete."""
    rng = np.random.default_rng(314)
    n, d, B, k = 200_000, 768, 256, 10
    raw = rng.standard_normal((n, d)).astype(np.float32)
    centre = rng.standard_normal(d).astype(np.float32)
    centre /= np.linalg.norm(centre)
    members = rng.choice(n, size=30_000, replace=False)
    raw[members] = centre + 0.12 * rng.standard_normal((30_000, d)).astype(np.float32) / np.sqrt(d)
    q = rng.standard_normal((B, d)).astype(np.float32)
    q[:200] = centre + 0.12 * rng.standard_normal((200, d)).astype(np.float32) / np.sqrt(d)
    ix = build(Index, raw)
    ix.set_option("shadow8_cooldown", 0)
    dist, rows = ix.search(q, k)
    assert ix.stat("i8v2_passes") == 1
    d_ref, i_ref = oracle_answer(raw, q, k, "f32")
    assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref)
    assert ix.stat("filter_hits") >= 200 * 25_000
    ix.close()


def test_tile_kernel_writes_the_same_candidate_lists_as_the_first_generation(Index):
    """Same sample, same anchors: the results must not depend on which kernel ran.  Since round 3 the tile kernel evaluates the
    int8 bound per 32-row block (each block's own quantisation error norm instead of the corpus's worst), so its candidate list
    is a SUBSET-sized one: fewer candidates than the first-generation kernel's (device-wide bound), never more than a
    hair above it (its integer per-value test has two accumulator levels of slack); `per_block` = 0 gives the old lists back."""
    rng = np.random.default_rng(77)
    n, d, B, k = 120_000, 768, 256, 10
    raw = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((B, d)).astype(np.float32)
    stats = []
    for v in (0, 1):
        ix = build(Index, raw, "f32", v)
        dist, rows = ix.search(q, k)
        assert ix.stat("i8v2_passes") == v
        stats.append((ix.stat("filter_hits"), ix.stat("filter_survivors"), dist.copy(), rows.copy()))
        ix.close()
    assert stats[0][0] * 0.4 <= stats[1][0] <= stats[0][0] * 1.01 + 8, (stats[0][0], stats[1][0])
    assert stats[1][1] <= stats[0][1] * 1.01 + 8, (stats[0][1], stats[1][1])
    assert np.array_equal(stats[0][2], stats[1][2]) and np.array_equal(stats[0][3], stats[1][3])
    # the device-wide bound in both kernels: the tile kernel's list holds the first-generation kernel's, plus a hair
    ix = build(Index, raw, "f32", 1)
    ix.set_option("per_block", 0)
    dist, rows = ix.search(q, k)
    assert stats[0][0] <= ix.stat("filter_hits") <= stats[0][0] * 1.01 + 8
    assert np.array_equal(dist, stats[0][2]) and np.array_equal(rows, stats[0][3])
    ix.close()


def test_tile_kernel_negative_and_zero_thresholds_and_zero_queries(Index):
    """Thresholds <= 0 (queries whose best sampled rows score below the slack), an all-zero query (scale 0: NaN or
    infinite threshold) and padding queries: the pre-test must hand all of them to the exact per-row test."""
    rng = np.random.default_rng(5)
    n, d, B, k = 30_000, 768, 140, 10
    raw = rng.standard_normal((n, d)).astype(np.float32)
    raw[:, :] *= (rng.random((n, 1)) < 0.5).astype(np.float32) * 0.999 + 0.001  # half of the rows nearly zero before normalisation (same direction: harmless)
    q = rng.standard_normal((B, d)).astype(np.float32)
    q[7] = 0.0
    q[8] = -raw[11]          # best score ~ -1 for row 11, thresholds of this query are low
    ix = build(Index, raw)
    dist, rows = ix.search(q, k)
    assert ix.stat("i8v2_passes") == 1
    d_ref, i_ref = oracle_answer(raw, q, k, "f32")
    assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref)
    ix.close()


def test_tile_kernel_dense_cluster_overflows_stay_exact(Index):
    """One tile of near-copies that 160 queries point at: the workgroup's LDS hit list fills inside one tile, the
    affected queries' counters are poisoned and they are answered by the next stage; everything stays exact."""
    rng = np.random.default_rng(6)
    n, d, B, k = 60_000, 768, 200, 10
    raw = rng.standard_normal((n, d)).astype(np.float32)
    centre = rng.standard_normal(d).astype(np.float32)
    raw[5120:5376] = centre + 1e-3 * rng.standard_normal((256, d)).astype(np.float32)   # exactly tile 20
    members = rng.choice(np.arange(6000, n), size=3000, replace=False)
    raw[members] = centre + 2e-3 * rng.standard_normal((3000, d)).astype(np.float32)    # and a spread-out cluster
    q = rng.standard_normal((B, d)).astype(np.float32)
    q[:160] = centre + 1e-3 * rng.standard_normal((160, d)).astype(np.float32)
    ix = build(Index, raw)
    dist, rows = ix.search(q, k)
    assert ix.stat("i8v2_passes") == 1
    d_ref, i_ref = oracle_answer(raw, q, k, "f32")
    assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref)
    ix.close()


def test_tile_kernel_repeated_searches_and_upserts(Index):
    """The same index searched repeatedly (stale LDS-DMA state, counters, cooldown logic must not leak between
    launches), with rows overwritten in between."""
    rng = np.random.default_rng(8)
    n, d, B, k = 50_000, 768, 256, 10
    raw = rng.standard_normal((n, d)).astype(np.float32)
    ix = build(Index, raw)
    for it in range(4):
        q = rng.standard_normal((B, d)).astype(np.float32)
        q[0] = raw[it * 1000 + 5]
        dist, rows = ix.search(q, k)
        d_ref, i_ref = oracle_answer(raw, q, k, "f32")
        assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref), it
        upd = rng.choice(n, size=500, replace=False)
        raw[upd] = rng.standard_normal((500, d)).astype(np.float32)
        ix.upsert(upd.astype(np.int64), raw[upd])
    assert ix.stat("i8v2_passes") == 4
    ix.close()


def test_forty_thousand_near_copies_stay_on_the_filter_path(Index):
    """The dense-cluster cliff of round 1: more near-identical rows than a query's candidate list held sent every query to
    32 sequential exact scans (168 ms instead of 2.7).  42,000 rows closer to each other than the int8 slack, 200 of 256
    queries pointing at them: the workgroup lists overflow inside every tile (direct global appends), every affected
    query collects 42,000 candidates, finalize re-scores them all — exact, and without the exact-scan fallback."""
    rng = np.random.default_rng(40)
    n, d, B, k = 60_000, 768, 256, 10
    raw = rng.standard_normal((n, d)).astype(np.float32)
    centre = rng.standard_normal(d).astype(np.float32)
    centre /= np.linalg.norm(centre)
    members = rng.choice(n, size=42_000, replace=False)
    raw[members] = centre + 0.1 * rng.standard_normal((42_000, d)).astype(np.float32) / np.sqrt(d)
    q = rng.standard_normal((B, d)).astype(np.float32)
    q[:200] = centre + 0.1 * rng.standard_normal((200, d)).astype(np.float32) / np.sqrt(d)
    ix = build(Index, raw)
    dist, rows = ix.search(q, k)
    assert ix.stat("i8v2_passes") == 1
    d_ref, i_ref = oracle_answer(raw, q, k, "f32")
    assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref)
    assert ix.stat("fallback_queries") == 0
    assert ix.stat("filter_survivors") >= 200 * 40_000
    ix.close()


@pytest.mark.parametrize(
    "shadow8,d,B",
    [
        (1, 768, 256),    # the static six-step program
        (0, 768, 256),    # ... its fp16 form (12 K-steps)
        (1, 768, 100),    # 8 query blocks
        (1, 384, 256),    # resident slices, one barrier per tile
        (1, 1024, 256),   # 8 K-steps: the generic interval loop
        (0, 384, 256),    # fp16, six K-steps
    ],
)
def test_candidate_lists_are_deterministic(Index, shadow8, d, B):
    """A race in the hand-ordered schedule (a stale slice, a load consumed before its counted wait) first shows as candidates that come
    and go between identical searches — long before it costs a true neighbour (round 3: the hazard of DESIGN.md 12.5, and a schedule
    variant that lost a few hundred of 12.8 million candidates per run while its top-10 still validated).  Ten identical searches
    through the int8 tile program (shadow8 = 1) and through its fp16 form (shadow8 = 0) must produce the same number of candidates
    and survivors every time."""
    rng = np.random.default_rng(99 + d + B)
    n, k = 200_000, 10
    raw = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((B, d)).astype(np.float32)
    ix = build(Index, raw, "f32", 2)
    ix.set_option("shadow8", shadow8)
    ix.set_option("shadow8_cooldown", 0)
    counts = []
    ref = None
    for rep in range(10):
        h0, s0 = ix.stat("filter_hits"), ix.stat("filter_survivors")
        dist, rows = ix.search(q, k)
        counts.append((ix.stat("filter_hits") - h0, ix.stat("filter_survivors") - s0))
        if ref is None:
            ref = (dist.copy(), rows.copy())
        assert np.array_equal(rows, ref[1]) and np.array_equal(dist, ref[0]), rep
    assert len(set(counts)) == 1, counts
    assert ix.stat("i8v2_passes" if shadow8 else "f16_tile_passes") == 10
    d_ref, i_ref = oracle_answer(raw, q, k, "f32")
    assert np.array_equal(ref[1], i_ref) and np.array_equal(ref[0], d_ref)
    ix.close()
