"""The int8 shadow ("shadow8", on by default for batches of <= 256 queries): half the bytes of the bf16 shadow per pass.  The filter is
still only a filter: survivors are re-scored with the canonical fp32 expression, so ids and distances must equal the
oracle's bit for bit; what changes is the error bound (per query, from the exact quantisation error norms)."""

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


def build8(Index, raw, dtype="f32"):
    ix = Index(raw.shape[1], dtype=dtype)
    ix.upsert(np.arange(raw.shape[0], dtype=np.int64), raw)
    for key in ("filter_min_rows", "filter_min_rows_small", "filter_min_batch"):
        ix.set_option(key, 1)
    ix.set_option("shadow8", 1)
    ix.set_option("shadow8_max_batch", 32)
    ix.set_option("small_batch_max", 0)   # (these tests are about the filter chain; the single-launch kernel has tests/test_gpu_small_batch.py)
    return ix


def quantise_like_the_kernels(x, is_query):
    """fp32 arithmetic of shadow8_from_rows_kernel / prep_queries8_kernel.  Stored rows share ONE scale per 32-row block (the
    block's largest magnitude / 127); a query has its own."""
    x = x.astype(np.float32)
    vmax = np.abs(x).max(axis=1).astype(np.float32)
    if not is_query:
        pad = (-len(vmax)) % 32
        vmax = np.repeat(np.concatenate([vmax, np.zeros(pad, np.float32)]).reshape(-1, 32).max(axis=1), 32)[: x.shape[0]]
    safe = np.where(vmax > 0, vmax, np.float32(1.0)).astype(np.float32)
    if is_query:
        scale = np.where(vmax > 0, vmax / np.float32(127.0), np.float32(0.0)).astype(np.float32)
        inv = np.where(vmax > 0, np.float32(127.0) / safe, np.float32(0.0)).astype(np.float32)
    else:
        scale = np.where(vmax > 0, vmax / np.float32(127.0), np.float32(1.0)).astype(np.float32)
        inv = (np.float32(1.0) / scale).astype(np.float32)
    q8 = np.clip(np.rint((x * inv[:, None]).astype(np.float32)), -127, 127).astype(np.int32)
    return q8, scale


@pytest.mark.parametrize("n,d,B,dtype", [(1000, 768, 20, "f32"), (300, 100, 32, "f32"), (700, 384, 7, "bf16"), (513, 1024, 1, "f16")])
def test_int8_scores_match_integer_reference(Index, n, d, B, dtype):
    """Operand layouts and scales: every int8 filter score equals (integer dot of the quantised operands) * row scale *
    query scale, evaluated with the kernel's own fp32 operations — bit for bit."""
    rng = np.random.default_rng(n + d)
    raw = rng.standard_normal((n, d)).astype(np.float32)
    raw[:, 0] += 3.0  # asymmetric data: a transposed or permuted operand cannot pass
    q = rng.standard_normal((B, d)).astype(np.float32)
    ix = build8(Index, raw, dtype)
    got = ix.approx_scores(q).cpu().numpy()
    assert ix.stat("shadow8_builds") == 1
    stored = o.widen(o.to_storage(o.normalize_rows(raw), dtype), dtype)
    c8, cs = quantise_like_the_kernels(stored, False)
    q8, qs = quantise_like_the_kernels(o.normalize_rows(q), True)
    acc = (q8.astype(np.int64) @ c8.astype(np.int64).T).astype(np.float32)  # |acc| < 2^24: exact in fp32
    ref = ((acc * cs[None, :]).astype(np.float32) * qs[:, None]).astype(np.float32)
    assert got.shape == (256, n)
    assert np.array_equal(got[:B], ref)
    assert not got[B:].any()
    # and the scores are what they claim to approximate, within the bound the filter uses
    exact = o.normalize_rows(q).astype(np.float64) @ stored.astype(np.float64).T
    e_r = np.linalg.norm(stored - c8 * cs[:, None], axis=1).max()
    e_q = np.linalg.norm(o.normalize_rows(q) - q8 * qs[:, None], axis=1)
    assert (np.abs(got[:B] - exact).max(axis=1) <= e_q * (1.01 + e_r) + 1.001 * e_r + 2e-6).all()
    ix.close()


@pytest.mark.parametrize(
    "n,d,B,k,dtype",
    [
        (70_000, 768, 1, 10, "f32"),     # the single-query latency path
        (70_000, 768, 32, 10, "f32"),
        (33_000, 768, 8, 10, "f32"),
        (20_001, 384, 17, 5, "f32"),     # ragged last tile, ragged batch
        (60_000, 200, 5, 100, "f32"),    # d not a multiple of 128 (dpad 256), k = 100
        (40_000, 768, 24, 10, "bf16"),
        (12_000, 1024, 3, 10, "f16"),
        (30_000, 64, 9, 10, "f32"),      # one 64-wide row padded to one 128-element K-step
        (30_000, 320, 31, 10, "f32"),    # 3 int8 K-steps (dpad8 = 384)
        (70_000, 768, 256, 10, "f32"),   # the headline shape, scaled down (NBQ = 8)
        (50_000, 768, 100, 10, "f32"),   # NBQ = 4
        (33_000, 384, 64, 10, "f32"),    # NBQ = 2
        (70_000, 256, 129, 100, "f32"),  # k = 100 (two list slots), ragged batch
        (40_000, 768, 96, 10, "bf16"),
        (20_000, 2048, 40, 10, "bf16"),  # widest row
        (50_000, 768, 300, 10, "f32"),   # two query passes (256 + 44), each with its own int8 query block
        (30_000, 128, 1024, 10, "f32"),  # the ABI's largest batch: four passes
    ],
)
def test_int8_filter_path_is_exact(Index, n, d, B, k, dtype):
    rng = np.random.default_rng(n + B)
    raw = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((B, d)).astype(np.float32)
    ix = build8(Index, raw, dtype)
    ix.set_option("shadow8_max_batch", 256)
    dist, rows = ix.search(q, k)
    assert ix.stat("shadow8_passes") == (B + 255) // 256 and ix.stat("shadow8_builds") == 1
    rows_ref = o.to_storage(o.normalize_rows(raw), dtype)
    d_ref, i_ref = o.search(rows_ref, dtype, o.normalize_rows(q), k)
    assert np.array_equal(rows, i_ref)
    assert np.array_equal(dist, d_ref)
    assert ix.stat("fallback_queries") == 0, "random data must not need the fallback"
    ix.close()


def test_int8_shadow_follows_upserts_and_larger_batches_keep_the_bf16_filter(Index):
    rng = np.random.default_rng(5)
    raw = rng.standard_normal((30_000, 256)).astype(np.float32)
    q = rng.standard_normal((4, 256)).astype(np.float32)
    ix = build8(Index, raw)
    ix.search(q, 10)
    assert ix.stat("shadow8_builds") == 1
    ix.search(q, 10)
    assert ix.stat("shadow8_builds") == 1  # unchanged rows: no rebuild
    # overwrite a row with the first query itself and append rows: the next search must see both
    more = rng.standard_normal((5_000, 256)).astype(np.float32)
    ix.upsert(np.array([123], dtype=np.int64), q[:1])
    ix.upsert(np.arange(30_000, 35_000, dtype=np.int64), more)
    raw[123] = q[0]
    raw = np.concatenate([raw, more])
    dist, rows = ix.search(q, 10)
    assert ix.stat("shadow8_builds") == 2
    assert rows[0, 0] == 123 and dist[0, 0] <= 2e-7
    d_ref, i_ref = o.search(o.normalize_rows(raw), "f32", o.normalize_rows(q), 10)
    assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref)
    # a batch above shadow8_max_batch takes the bf16 filter
    ix.set_option("shadow8_max_batch", 8)
    passes8 = ix.stat("shadow8_passes")
    q_big = rng.standard_normal((40, 256)).astype(np.float32)
    dist, rows = ix.search(q_big, 10)
    assert ix.stat("shadow8_passes") == passes8
    d_ref, i_ref = o.search(o.normalize_rows(raw), "f32", o.normalize_rows(q_big), 10)
    assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref)
    ix.close()


def test_int8_zero_query_duplicates_and_clusters_stay_exact(Index):
    """A zero query (scale 0), exact duplicates (ties -> lower row) and a dense cluster (many candidates inside the
    int8 slack) must all come out as the oracle says — through the fallback where the candidate lists overflow."""
    rng = np.random.default_rng(6)
    raw = rng.standard_normal((40_000, 128)).astype(np.float32)
    centre = rng.standard_normal(128).astype(np.float32)
    raw[1000:3000] = centre + 0.01 * rng.standard_normal((2000, 128)).astype(np.float32)
    raw[20_000] = raw[7]
    q = np.stack([np.zeros(128, dtype=np.float32), raw[7], centre, rng.standard_normal(128).astype(np.float32)])
    ix = build8(Index, raw)
    dist, rows = ix.search(q, 10)
    d_ref, i_ref = o.search(o.normalize_rows(raw), "f32", o.normalize_rows(q), 10)
    assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref)
    assert rows[1, 0] == 7 and rows[1, 1] == 20_000
    ix.close()


def test_badly_quantisable_rows_turn_the_int8_filter_off_not_wrong(Index):
    """One large element and many small ones: int8 rounds the small ones away, the worst-row error norm is large and the
    int8 bound useless.  The first search still answers exactly (through the fallback if need be); once the error
    norm has been read back the index keeps the bf16 filter for small batches."""
    import torch

    rng = np.random.default_rng(8)
    raw = rng.standard_normal((30_000, 512)).astype(np.float32)
    raw[:2000, 0] = 40.0  # after normalisation: one element ~0.87, the others ~0.02 = 2-3 int8 steps
    q = rng.standard_normal((3, 512)).astype(np.float32)
    ix = build8(Index, raw)
    d_ref, i_ref = o.search(o.normalize_rows(raw), "f32", o.normalize_rows(q), 10)
    dist, rows = ix.search(q, 10)
    assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref)
    torch.cuda.synchronize()
    assert ix.stat("shadow8_eps_r_micro") > 40_000
    passes8 = ix.stat("shadow8_passes")
    dist, rows = ix.search(q, 10)
    assert ix.stat("shadow8_passes") == passes8, "the index should have gone back to the bf16 filter"
    assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref)
    ix.close()


def test_dense_clusters_send_the_index_back_to_the_bf16_filter(Index):
    """Clustered corpora (what embedding corpora look like): the int8 slack lets whole clusters through to the exact
    re-scoring.  The index watches its own counters and takes the bf16 filter for a while; results are the same bits
    either way."""
    import torch

    n, d, B, k = 200_000, 256, 64, 10
    g = torch.Generator(device="cuda").manual_seed(3)
    c = torch.nn.functional.normalize(torch.randn((16, d), generator=g, device="cuda"), dim=1)
    x = c[torch.randint(0, 16, (n,), generator=g, device="cuda")] + 0.3 * torch.randn((n, d), generator=g, device="cuda") / d ** 0.5
    q = c[torch.randint(0, 16, (B,), generator=g, device="cuda")] + 0.3 * torch.randn((B, d), generator=g, device="cuda") / d ** 0.5
    ix = Index(d)
    ix.upsert_device(0, x.contiguous())
    outs = []
    for _ in range(4):
        outs.append(ix.search_tensors(q, k))
        torch.cuda.synchronize()
    assert ix.stat("shadow8_passes") >= 1 and ix.stat("shadow8_cooldowns") >= 1
    passes8 = ix.stat("shadow8_passes")
    outs.append(ix.search_tensors(q, k))
    torch.cuda.synchronize()
    assert ix.stat("shadow8_passes") == passes8, "cooling down: this search must have taken the bf16 filter"
    ix.set_option("filter", 0)
    d_ref, r_ref = ix.search_tensors(q, k)
    for dd, rr in outs:
        assert torch.equal(rr, r_ref) and torch.equal(dd, d_ref)
    ix.close()


def test_rebuild_on_one_stream_is_awaited_by_searches_on_others(Index):
    """After rows changed, the first search rebuilds the int8 shadow on ITS stream; a search issued right behind it on
    another stream must wait for that build on the device, not read a half-written shadow."""
    import torch

    rng = np.random.default_rng(9)
    raw = rng.standard_normal((150_000, 256)).astype(np.float32)
    ix = build8(Index, raw)
    qs = [torch.from_numpy(rng.standard_normal((b, 256)).astype(np.float32)).cuda() for b in (4, 30)]
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for round_ in range(3):
        new = rng.standard_normal((20_000, 256)).astype(np.float32)
        first = 150_000 + 20_000 * round_
        ix.upsert(np.arange(first, first + 20_000, dtype=np.int64), new)     # rows changed: the shadow is stale
        raw = np.concatenate([raw, new])
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            o1 = ix.search_tensors(qs[0], 10)                                  # rebuilds on s1
        with torch.cuda.stream(s2):
            o2 = ix.search_tensors(qs[1], 10)                                  # must wait for it on s2
        torch.cuda.synchronize()
        rows_ref = o.normalize_rows(raw)
        for q, (dd, rr) in zip(qs, (o1, o2)):
            d_ref, i_ref = o.search(rows_ref, "f32", o.normalize_rows(q.cpu().numpy()), 10)
            assert np.array_equal(rr.cpu().numpy(), i_ref) and np.array_equal(dd.cpu().numpy(), d_ref)
    assert ix.stat("shadow8_builds") == 3  # one per round: the first one builds everything, the others only the appended rows
    ix.close()


@pytest.mark.parametrize(
    "n,d,B,k,dtype",
    [
        (70_000, 384, 256, 10, "f32"),   # MiniLM width: 3 int8 K-steps, the whole query block stays in LDS
        (50_000, 512, 100, 10, "f32"),   # 4 K-steps: all four LDS slices hold a slice of their own
        (40_000, 256, 32, 10, "bf16"),   # 2 K-steps
        (30_000, 100, 7, 10, "f32"),     # 1 K-step (dpad8 = 128), ragged batch
        (60_001, 384, 1, 100, "f16"),    # ragged last tile, single query, k = 100 (needs 2k sample tiles)
    ],
)
def test_resident_query_block_is_exact(Index, n, d, B, k, dtype):
    """Rows of <= 512 int8 elements: the query block is loaded into LDS once per workgroup and never re-staged
    (gemm_filter_kernel<., ., 1, RES=1>).  Same bits as the oracle, and as the re-staging kernel."""
    rng = np.random.default_rng(n + d + B)
    raw = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((B, d)).astype(np.float32)
    ix = build8(Index, raw, dtype)
    ix.set_option("shadow8_max_batch", 256)
    dist, rows = ix.search(q, k)
    assert ix.stat("shadow8_passes") == 1 and ix.stat("fallback_queries") == 0
    rows_ref = o.to_storage(o.normalize_rows(raw), dtype)
    d_ref, i_ref = o.search(rows_ref, dtype, o.normalize_rows(q), k)
    assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref)
    ix.set_option("resident_q", 0)
    dist2, rows2 = ix.search(q, k)
    assert np.array_equal(rows2, rows) and np.array_equal(dist2, dist)
    ix.close()


def test_six_step_rows_with_two_resident_slices_are_exact(Index):
    """768 int8 elements, 65..128 queries: gemm_filter_kernel<., 4, 1, RES=2> keeps two of the six query slices in LDS and
    rings the other four; same bits as the oracle and as the re-staging kernel."""
    rng = np.random.default_rng(12)
    raw = rng.standard_normal((80_000, 768)).astype(np.float32)
    ix = build8(Index, raw)
    ix.set_option("shadow8_max_batch", 256)
    rows_ref = o.normalize_rows(raw)
    for B in (65, 100, 128):
        q = rng.standard_normal((B, 768)).astype(np.float32)
        dist, rows = ix.search(q, 10)
        d_ref, i_ref = o.search(rows_ref, "f32", o.normalize_rows(q), 10)
        assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref)
        ix.set_option("resident_q", 0)
        dist2, rows2 = ix.search(q, 10)
        ix.set_option("resident_q", 1)
        assert np.array_equal(rows2, rows) and np.array_equal(dist2, dist)
    assert ix.stat("fallback_queries") == 0
    ix.close()


def test_one_badly_quantising_row_no_longer_widens_every_query(Index):
    """the int8 bound's row-error term was ONE device-wide running maximum — a single badly quantising row
    widened every query's slack (or, above 0.04, sent the whole index to the slower 2-byte filter) for the life of the index,
    even after the row was overwritten.  Round 3: i8_tile_kernel and finalize evaluate the bound per 32-row block, the
    device-wide figure is re-derived from the blocks after every build, and a few wide blocks do not switch the filter off."""
    import torch

    rng = np.random.default_rng(21)
    n, d, B, k = 90_000, 768, 256, 10
    raw = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((B, d)).astype(np.float32)

    def fresh(rows):
        ix = Index(d)
        ix.upsert(np.arange(rows.shape[0], dtype=np.int64), rows)
        for key in ("filter_min_rows", "filter_min_rows_small", "filter_min_batch"):
            ix.set_option(key, 1)
        return ix

    ix = fresh(raw)
    ix.search(q, k)
    base_hits = ix.stat("filter_hits")
    base_eps = ix.stat("shadow8_eps_r_micro")
    assert 0 < base_eps < 40_000 and ix.stat("i8v2_passes") == 1
    ix.close()

    bad = raw.copy()
    bad[[5, 40_000, 77_777], 0] = 300.0     # three rows of one large element and many tiny ones: int8 rounds the tiny ones away
    ix = fresh(bad)
    d_ref, i_ref = o.search(o.normalize_rows(bad), "f32", o.normalize_rows(q), k)
    dist, rows = ix.search(q, k)             # (first search: the error norms have not been read back yet)
    assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref)
    torch.cuda.synchronize()
    assert ix.stat("shadow8_eps_r_micro") > 40_000 and ix.stat("shadow8_wide_blocks") == 3
    passes8 = ix.stat("shadow8_passes")
    h0 = ix.stat("filter_hits")
    dist, rows = ix.search(q, k)
    assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref)
    assert ix.stat("shadow8_passes") == passes8 + 1, "three wide blocks in 2,800 must not switch the int8 filter off for this batch size"
    # the three wide blocks contribute at most their own rows as extra candidates (3 x 32 per query); nobody else's slack moved
    assert ix.stat("filter_hits") - h0 <= base_hits * 1.02 + 3 * 32 * B
    # the rows are overwritten with ordinary ones: the device-wide figure comes back down (it used to stay up for ever)
    ix.upsert(np.array([5, 40_000, 77_777], dtype=np.int64), raw[[5, 40_000, 77_777]])
    ix.search(q, k)
    torch.cuda.synchronize()
    ix.search(q[:8].copy(), k)               # (the read-back is looked at by the next search)
    assert ix.stat("shadow8_eps_r_micro") < 40_000 and ix.stat("shadow8_wide_blocks") == 0
    d2, r2 = ix.search(q, k)
    d_ref2, i_ref2 = o.search(o.normalize_rows(raw), "f32", o.normalize_rows(q), k)
    assert np.array_equal(r2, i_ref2) and np.array_equal(d2, d_ref2)
    ix.close()
