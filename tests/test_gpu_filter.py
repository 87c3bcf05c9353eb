"""GPU parity of the large-batch leg: bf16 MFMA filter + exact fp32 re-score (csrc/filter_gemm.h).

The filter is only allowed to change SPEED: final ids and distances must stay bit-identical to
the canonical CPU oracle, including when its candidate lists overflow and the exact-scan
fallback takes over."""

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


@pytest.fixture(scope="module")
def torch():
    import torch

    return torch


def build(Index, raw, dtype="f32", force_filter=True):
    ix = Index(raw.shape[1], dtype=dtype)
    ix.upsert(np.arange(raw.shape[0], dtype=np.int64), raw)
    ix.set_option("shadow8", 0)  # this file is about the bf16 filter; the int8 one (batches <= 8) has tests/test_gpu_shadow8.py
    if force_filter:
        ix.set_option("filter_min_rows", 1)
        ix.set_option("filter_min_rows_small", 1)
        ix.set_option("filter_min_batch", 1)
    return ix


def oracle_answer(raw, q_raw, k, dtype):
    rows = o.to_storage(o.normalize_rows(raw), dtype)
    return o.search(rows, dtype, o.normalize_rows(q_raw), k)


@pytest.mark.parametrize("n,d,B,dtype", [(1000, 768, 40, "f32"), (300, 128, 256, "f32"), (700, 384, 7, "bf16"), (513, 1024, 33, "f16")])
def test_mfma_scores_match_rounded_operand_reference(Index, n, d, B, dtype):
    """Operand layouts: every approximate score equals the dot product of the stored row and the query,
    both rounded to the shadow element type (bf16, or fp16 in a CODD_SHADOW_F16 build), fp32
    accumulation order aside."""
    rng = np.random.default_rng(n + d)
    raw = rng.standard_normal((n, d)).astype(np.float32)
    raw[:, 0] += 3.0  # asymmetric data: a transposed or permuted operand cannot pass
    q = rng.standard_normal((B, d)).astype(np.float32)
    ix = build(Index, raw, dtype)
    got = ix.approx_scores(q).cpu().numpy()
    from codd_query_engine_amd import native

    shadow = "f16" if "shadow=f16" in native.load().codd_knn_version().decode() else "bf16"
    stored = o.widen(o.to_storage(o.normalize_rows(raw), dtype), dtype)
    cb = o.widen(o.to_storage(stored, shadow), shadow).astype(np.float64)
    qb = o.widen(o.to_storage(o.normalize_rows(q), shadow), shadow).astype(np.float64)
    ref = qb @ cb.T
    assert got.shape == (256, n)
    assert np.abs(got[:B] - ref).max() < 2e-5
    assert not got[B:].any()
    ix.close()


@pytest.mark.parametrize(
    "n,d,B,k,dtype",
    [
        (70_000, 768, 256, 10, "f32"),   # the headline shape, scaled down
        (33_000, 768, 64, 10, "f32"),
        (20_001, 384, 17, 5, "f32"),     # ragged last tile, ragged batch
        (50_000, 768, 300, 10, "f32"),   # two query passes (256 + 44)
        (60_000, 256, 32, 100, "f32"),   # k = 100 (two list slots)
        (40_000, 768, 96, 10, "bf16"),
        (12_000, 1024, 48, 10, "f16"),
        (5_200, 128, 20, 10, "f32"),     # 21 tiles: barely enough buckets for k = 10
        (30_000, 64, 40, 10, "f32"),     # ONE K-step per tile (query stages wrap inside a stage)
        (30_000, 320, 40, 10, "f32"),    # odd number of K-steps (5): tiles end mid-stage
        (8_000, 2048, 24, 10, "bf16"),   # widest row (32 K-steps)
        (30_000, 128, 1024, 10, "f32"),  # the ABI's largest batch: four query passes
        (70_000, 256, 16, 128, "f32"),   # the ABI's largest k (needs 2k = 256 sample tiles)
    ],
)
def test_filter_path_is_exact(Index, n, d, B, k, dtype):
    rng = np.random.default_rng(n + B)
    raw = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((B, d)).astype(np.float32)
    ix = build(Index, raw, dtype)
    dist, rows = ix.search(q, k)
    assert ix.stat("filter_passes") == (B + 255) // 256
    d_ref, i_ref = oracle_answer(raw, q, k, dtype)
    assert np.array_equal(rows, i_ref)
    assert np.array_equal(dist, d_ref)
    assert ix.stat("fallback_queries") == 0, "random data must not need the fallback"
    assert ix.stat("filter_survivors") < ix.stat("filter_hits") <= B * 8192
    ix.close()


@pytest.mark.parametrize("B", [1, 5, 32, 33, 64, 100, 128, 129])
def test_every_query_block_count_is_exact(Index, B):
    """The GEMM is instantiated for 1, 2, 4 and 8 blocks of 32 queries; batch sizes on both sides of
    every boundary must give the oracle's bits (B = 1 is the single-query latency path on large corpora)."""
    rng = np.random.default_rng(100 + B)
    n, d, k = 30_000, 768, 10
    raw = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((B, d)).astype(np.float32)
    ix = build(Index, raw)
    dist, rows = ix.search(q, k)
    assert ix.stat("filter_passes") == 1 and ix.stat("fallback_queries") == 0
    d_ref, i_ref = oracle_answer(raw, q, k, "f32")
    assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref)
    ix.close()


def test_path_selection_follows_the_measured_rule(Index):
    """<= 8 queries: exact scan until rows * B reaches 100k; more than 8 queries: the filter whenever the
    corpus has enough tiles for sound thresholds (scripts/crossover.py is where the rule was measured)."""
    rng = np.random.default_rng(4)
    raw = rng.standard_normal((40_000, 128)).astype(np.float32)
    ix = Index(128)
    ix.upsert(np.arange(40_000, dtype=np.int64), raw)
    ix.search(rng.standard_normal((2, 128)).astype(np.float32), 5)      # 80k < 100k: scan
    assert ix.stat("filter_passes") == 0
    ix.search(rng.standard_normal((3, 128)).astype(np.float32), 5)      # 120k: filter
    assert ix.stat("filter_passes") == 1
    ix.search(rng.standard_normal((16, 128)).astype(np.float32), 5)     # > 8 queries: filter
    assert ix.stat("filter_passes") == 2
    small = Index(128)
    small.upsert(np.arange(3_000, dtype=np.int64), raw[:3_000])         # 12 tiles < 2k: never sound for k = 10
    d, r = small.search(rng.standard_normal((64, 128)).astype(np.float32), 10)
    assert small.stat("filter_passes") == 0 and small.stat("scan_launches") == 1   # one launch for the whole batch
    ix.close(); small.close()


def test_overflow_falls_back_and_stays_exact(Index):
    """5,000 near-copies of one vector: every one of them clears any sound threshold, the
    per-query candidate list (shrunk to 64 here) overflows, and the exact scan must take over
    for the affected queries only."""
    rng = np.random.default_rng(1)
    n, d, B, k = 40_000, 256, 24, 10
    raw = rng.standard_normal((n, d)).astype(np.float32)
    centre = rng.standard_normal(d).astype(np.float32)
    dup = rng.choice(n, size=5000, replace=False)
    raw[dup] = centre + 1e-3 * rng.standard_normal((5000, d)).astype(np.float32)
    q = rng.standard_normal((B, d)).astype(np.float32)
    q[3] = centre
    q[11] = centre + 1e-3 * rng.standard_normal(d).astype(np.float32)
    ix = build(Index, raw)
    ix.set_option("hit_cap", 64)
    dist, rows = ix.search(q, k)
    d_ref, i_ref = oracle_answer(raw, q, k, "f32")
    assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref)
    assert 2 <= ix.stat("fallback_queries") < B
    ix.close()


@pytest.mark.parametrize("B,d,dtype,fuse", [(200, 768, "f32", 1), (200, 768, "f32", 0), (96, 256, "bf16", 1), (130, 1024, "f16", 1),
                                            (200, 768, "bf16", 1)])   # (the last one: the 2-byte filter through the tile program on fp16 operands)
def test_overflow_in_a_large_batch_is_answered_inside_the_finalize_launch(Index, B, d, dtype, fuse):
    """Round 3: for batches above 64 queries finalize and the exact-scan fallback are ONE launch (finalize_fb_kernel: its scan
    workgroups derive the queue from the hit counters).  Same overflow as above — 5,000 near-copies, candidate lists shrunk to
    64 — on the int8 tile kernel's and the first-generation kernels' batch sizes and on 2-byte rows; "fuse_fallback" = 0 is
    the two-launch form.  Affected queries only; same bits as the oracle; repeated searches reuse the arrival ticket."""
    rng = np.random.default_rng(B + d)
    n, k = 40_000, 10
    raw = rng.standard_normal((n, d)).astype(np.float32)
    centre = rng.standard_normal(d).astype(np.float32)
    dup = rng.choice(n, size=5000, replace=False)
    raw[dup] = centre + 1e-3 * rng.standard_normal((5000, d)).astype(np.float32)
    q = rng.standard_normal((B, d)).astype(np.float32)
    hot = [3, 11, 77, B - 1]
    for h in hot:
        q[h] = centre + 1e-3 * rng.standard_normal(d).astype(np.float32)
    ix = build(Index, raw, dtype)
    ix.set_option("shadow8", 1 if dtype == "f32" else 0)   # (f32 cases: the int8 tile kernel's pass; 2-byte rows: the fp16 filter's)
    ix.set_option("hit_cap", 1024 if dtype == "f32" else 64)   # (the int8 slack leaves an ordinary query ~150 candidates here, the fp16 one a dozen)
    ix.set_option("fuse_fallback", fuse)
    ix.set_option("shadow8_cooldown", 0)
    d_ref, i_ref = oracle_answer(raw, q, k, dtype)
    for rep in range(2):
        dist, rows = ix.search(q, k)
        assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref), rep
        # (the hot queries, and at most a couple of ordinary ones whose sample happened to anchor a weak threshold)
        assert (rep + 1) * len(hot) <= ix.stat("fallback_queries") <= (rep + 1) * (len(hot) + 2)
    ix.close()


def test_workgroup_hit_list_overflow_goes_straight_to_the_global_lists(Index):
    """One tile holds 256 near-copies of a vector and 24 queries point at it: 6,144 hits land in one workgroup's
    2,048-entry LDS list inside a single tile.  Round 1 poisoned the affected queries' counters and sent them to the
    exact scan; now the overflow is appended to the queries' global lists directly: every answer exact, no fallback."""
    rng = np.random.default_rng(6)
    n, d, B, k = 36_000, 256, 40, 10
    raw = rng.standard_normal((n, d)).astype(np.float32)
    centre = rng.standard_normal(d).astype(np.float32)
    raw[5120:5376] = centre + 1e-3 * rng.standard_normal((256, d)).astype(np.float32)  # exactly tile 20
    q = rng.standard_normal((B, d)).astype(np.float32)
    q[:24] = centre + 1e-3 * rng.standard_normal((24, d)).astype(np.float32)
    ix = build(Index, raw)
    dist, rows = ix.search(q, k)
    d_ref, i_ref = oracle_answer(raw, q, k, "f32")
    assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref)
    assert ix.stat("fallback_queries") == 0
    assert ix.stat("filter_hits") >= 24 * 256
    ix.close()


@pytest.mark.parametrize("shadow8,B", [(0, 256), (1, 100)])
def test_forty_thousand_near_copies_first_generation_kernels(Index, shadow8, B):
    """The dense-cluster cliff on the kernels of filter_gemm.h (the bf16 filter; the int8 filter of batches up to 128
    queries): 42,000 rows closer to each other than the filter's slack.  Every affected query keeps all of its
    candidates (131,072 per query by default) and is answered without the exact-scan fallback."""
    rng = np.random.default_rng(41)
    n, d, k = 60_000, 768, 10
    raw = rng.standard_normal((n, d)).astype(np.float32)
    centre = rng.standard_normal(d).astype(np.float32)
    centre /= np.linalg.norm(centre)
    members = rng.choice(n, size=42_000, replace=False)
    # (spread of the cluster: inside the slack of the filter under test — the 2-byte shadow is fp16 since round 3, eps 0.0011
    # at 768-d, seven times tighter than bf16's: the same cluster needs a fifth of the noise to stay inseparable)
    noise = 0.1 if shadow8 else 0.02
    raw[members] = centre + noise * rng.standard_normal((42_000, d)).astype(np.float32) / np.sqrt(d)
    q = rng.standard_normal((B, d)).astype(np.float32)
    hot = B * 3 // 4
    q[:hot] = centre + noise * rng.standard_normal((hot, d)).astype(np.float32) / np.sqrt(d)
    ix = build(Index, raw)
    ix.set_option("shadow8", shadow8)
    dist, rows = ix.search(q, k)
    d_ref, i_ref = oracle_answer(raw, q, k, "f32")
    assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref)
    assert ix.stat("fallback_queries") == 0
    assert ix.stat("filter_survivors") >= hot * 40_000
    ix.close()


def test_dense_cluster_costs_time_not_exactness(Index):
    """3,000 rows within a hair of each other and of the query: all of them survive the 2-eps band, more than
    one finalize round (2,048 hits) can hold; the rounds stream through them, no cap, no fallback."""
    rng = np.random.default_rng(16)
    n, d, k = 30_000, 256, 10
    raw = rng.standard_normal((n, d)).astype(np.float32)
    centre = rng.standard_normal(d).astype(np.float32)
    members = rng.choice(n, size=3000, replace=False)
    raw[members] = centre + 2e-3 * rng.standard_normal((3000, d)).astype(np.float32)
    q = rng.standard_normal((10, d)).astype(np.float32)
    q[0] = centre
    ix = build(Index, raw)
    dist, rows = ix.search(q, k)
    d_ref, i_ref = oracle_answer(raw, q, k, "f32")
    assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref)
    assert ix.stat("fallback_queries") == 0 and ix.stat("filter_survivors") >= 3000
    ix.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_few_valued_rows_do_not_slip_under_the_bf16_bound(Index, dtype):
    """Rows whose entries all carry the SAME rounding error are the worst case of the bf16 bound: a unit vector of
    239 equal entries scores 0.99286 against itself in bf16 x bf16 (error 0.0071).  With the unit roundoff taken as
    2^-9 (round 1) the slack was 0.0040 and such a row — the true top-1 — fell under a threshold anchored on a
    sampled near-duplicate.  Few-valued vectors are what the hashing embedder produces."""
    rng = np.random.default_rng(239)
    n, d, k = 30_000, 768, 10
    raw = rng.standard_normal((n, d)).astype(np.float32)
    base = np.zeros(d, dtype=np.float32)
    base[:239] = 1.0
    members = np.sort(rng.choice(n, size=400, replace=False))
    for j, r in enumerate(members):
        row = base.copy()
        z = j % 20                      # cosine with `base` = sqrt((239 - z) / 239): 1.0 down to 0.959
        row[rng.choice(239, size=z, replace=False)] = 0.0
        raw[r] = row
    two = base.copy()
    two[:120] = 0.35                   # the embedder's two magnitudes
    raw[members[-40:]] = two
    q = rng.standard_normal((24, d)).astype(np.float32)
    q[0] = base
    q[1] = two
    q[2] = base * 3.0
    q[3, :239] = 1.0
    ix = build(Index, raw, dtype)
    dist, rows = ix.search(q, k)
    assert ix.stat("filter_passes") == 1
    d_ref, i_ref = oracle_answer(raw, q, k, dtype)
    assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref)
    ix.close()


def test_zero_query_and_zero_rows_through_the_filter(Index):
    """A zero query scores 0 against everything: every row is a candidate and a survivor (20,000 of them, streamed
    through finalize in rounds) and the answer is rows 0..k-1 at distance exactly 1; zero rows score 0 and never
    beat real neighbours."""
    rng = np.random.default_rng(12)
    n, d, k = 20_000, 256, 10
    raw = rng.standard_normal((n, d)).astype(np.float32)
    raw[100:110] = 0.0
    q = rng.standard_normal((12, d)).astype(np.float32)
    q[5] = 0.0
    ix = build(Index, raw)
    dist, rows = ix.search(q, k)
    d_ref, i_ref = oracle_answer(raw, q, k, "f32")
    assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref)
    assert rows[5].tolist() == list(range(k)) and (dist[5] == 1.0).all()
    assert ix.stat("filter_survivors") >= n
    ix.set_option("hit_cap", 4096)      # now the zero query's 20,000 candidates overflow: exact-scan fallback, same bits
    dist2, rows2 = ix.search(q, k)
    assert np.array_equal(rows2, i_ref) and np.array_equal(dist2, d_ref) and ix.stat("fallback_queries") >= 1
    ix.close()


def test_ties_resolve_to_lower_row_through_the_filter(Index):
    rng = np.random.default_rng(2)
    n, d = 35_000, 768
    raw = rng.standard_normal((n, d)).astype(np.float32)
    for r in (17, 9000, 20_000, 34_999):
        raw[r] = raw[5]
    q = rng.standard_normal((16, d)).astype(np.float32)
    q[0] = raw[5]
    ix = build(Index, raw)
    dist, rows = ix.search(q, 10)
    assert rows[0, :5].tolist() == [5, 17, 9000, 20_000, 34_999]
    d_ref, i_ref = oracle_answer(raw, q, 10, "f32")
    assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref)
    ix.close()


def test_filter_and_exact_scan_agree_at_scale(Index):
    """1M x 768, B=256: the two legs of the engine must return the same bits (the oracle cannot
    brute-force this size quickly; the exact scan is itself oracle-checked at smaller sizes)."""
    import torch

    n, d, B, k = 1_000_000, 768, 256, 10
    g = torch.Generator(device="cuda").manual_seed(7)
    ix = Index(d)
    ix.reserve(n)
    for c in range(4):
        ix.upsert_device(c * 250_000, torch.randn((250_000, d), generator=g, device="cuda"))
    q = torch.randn((B, d), generator=g, device="cuda")
    d_f, r_f = ix.search_tensors(q, k)
    assert ix.stat("filter_passes") == 1 and ix.stat("fallback_queries") == 0
    ix.set_option("filter", 0)
    d_e, r_e = ix.search_tensors(q, k)
    assert torch.equal(r_f, r_e) and torch.equal(d_f, d_e)
    ix.close()


def test_too_few_tiles_for_k_takes_the_exact_scan(Index):
    rng = np.random.default_rng(8)
    raw = rng.standard_normal((9_000, 256)).astype(np.float32)  # 36 tiles < 2 * 100
    q = rng.standard_normal((32, 256)).astype(np.float32)
    ix = build(Index, raw)
    dist, rows = ix.search(q, 100)
    assert ix.stat("filter_passes") == 0
    d_ref, i_ref = oracle_answer(raw, q, 100, "f32")
    assert np.array_equal(rows, i_ref) and np.array_equal(dist, d_ref)
    ix.close()


def test_full_baseline_size_properties(Index):
    """BASELINE configs[2] at full size (10M x 768 fp32, B = 256, k = 10), through size-independent properties:
    planted near-duplicates come back in planted order; distances ascend; the filter leg and the exact scan
    return the same bits for a slice of the batch; one query (B = 1 path) equals its row of the batch."""
    import torch

    n, d, B, k = 10_000_000, 768, 256, 10
    ix = Index(d)
    ix.reserve(n)
    for c in range(n // 250_000):
        g = torch.Generator(device="cuda").manual_seed(1234 + c)
        ix.upsert_device(c * 250_000, torch.randn((250_000, d), generator=g, device="cuda"))
    q = torch.randn((B, d), generator=torch.Generator(device="cuda").manual_seed(4321), device="cuda")
    gp = torch.Generator(device="cuda").manual_seed(99)
    planted = []
    for b in range(3):
        for j in range(k):
            row = (b * 1_234_567 + j * 99_991 + 17) % n
            noise = torch.randn(d, generator=gp, device="cuda")
            ix.upsert_device(row, (q[b] + 0.02 * (j + 1) * noise * q[b].norm() / noise.norm())[None, :].contiguous())
            planted.append(row)
    dist, rows = ix.search_tensors(q, k)
    assert rows[:3].cpu().numpy().tolist() == np.array(planted).reshape(3, k).tolist()
    assert bool((dist[:, 1:] >= dist[:, :-1]).all()) and bool((rows >= 0).all())
    assert ix.stat("filter_passes") == 1 and ix.stat("fallback_queries") == 0
    assert ix.stat("shadow8_passes") == 1                      # the default for a batch of <= 256 queries: the int8 filter
    d1, r1 = ix.search_tensors(q[7:8], k)                      # single-query path (NBQ = 1 instantiation)
    assert torch.equal(r1[0], rows[7]) and torch.equal(d1[0], dist[7])
    ix.set_option("shadow8", 0)                                # the bf16 filter must return the same bits
    d2, r2 = ix.search_tensors(q, k)
    assert ix.stat("shadow8_passes") == 2 and torch.equal(r2, rows) and torch.equal(d2, dist)
    ix.set_option("filter", 0)
    de, re_ = ix.search_tensors(q[:8], k)                      # exact scan, one pass over 30.7 GB
    assert torch.equal(re_, rows[:8]) and torch.equal(de, dist[:8])
    ix.close()


def test_unnormalised_rows_disable_the_filter(Index):
    rng = np.random.default_rng(3)
    raw = rng.standard_normal((40_000, 128)).astype(np.float32)
    ix = Index(128)
    ix.upsert(np.arange(40_000, dtype=np.int64), raw, normalize=False)  # caller's own scaling: no unit-norm bound
    ix.search(rng.standard_normal((32, 128)).astype(np.float32), 5)
    assert ix.stat("filter_passes") == 0
    ix.close()


def test_searches_on_several_streams_use_their_own_workspaces(Index, torch):
    """The next batch may be issued on another stream while the previous one is still in flight (bench.py's sharded
    loop does): every stream gets its own workspace, a fifth stream takes over the least recently used one."""
    rng = np.random.default_rng(77)
    raw = rng.standard_normal((60_000, 256)).astype(np.float32)
    ix = Index(256)
    ix.upsert(np.arange(raw.shape[0], dtype=np.int64), raw)
    qs = [torch.from_numpy(rng.standard_normal((b, 256)).astype(np.float32)).cuda() for b in (256, 40, 1, 130, 256, 9)]
    ref = [ix.search_tensors(q, 10) for q in qs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(6)]
    cur = torch.cuda.current_stream()
    outs = []
    for rep in range(5):
        for i, q in enumerate(qs):
            s = streams[(i + rep) % len(streams)]
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs.append((i, ix.search_tensors(q, 10)))
    for s in streams:
        cur.wait_stream(s)
    torch.cuda.synchronize()
    assert ix.stat("workspaces") == 4
    for i, (d, r) in outs:
        assert torch.equal(r, ref[i][1]) and torch.equal(d, ref[i][0])
    ix.close()


def test_search_async_pipelines_batches(Index, torch):
    from codd_query_engine_amd.sharded import ShardedSearcher

    rng = np.random.default_rng(78)
    raw = rng.standard_normal((30_000, 128)).astype(np.float32)
    ix = Index(128)
    ix.upsert(np.arange(raw.shape[0], dtype=np.int64), raw)
    searcher = ShardedSearcher(ix, row_base=0)
    qs = [torch.from_numpy(rng.standard_normal((64, 128)).astype(np.float32)).cuda() for _ in range(7)]
    ref = [searcher.search(q, 10) for q in qs]
    pending = [searcher.search_async(q, 10, depth=2) for q in qs]
    for h, (d_ref, r_ref) in zip(pending, ref):
        d, r = h.result()
        assert torch.equal(r, r_ref) and torch.equal(d, d_ref)
    ix.close()


def test_concurrent_host_threads_search_one_index(Index, torch):
    """The service shares one store between request handlers (metrics_controller.py:22-38): several host threads may
    search one index at once, on the default stream or on streams of their own."""
    import threading

    rng = np.random.default_rng(79)
    raw = rng.standard_normal((40_000, 192)).astype(np.float32)
    ix = Index(192)
    ix.upsert(np.arange(raw.shape[0], dtype=np.int64), raw)
    qs = [torch.from_numpy(rng.standard_normal((b, 192)).astype(np.float32)).cuda() for b in (1, 17, 64, 200)]
    ref = [ix.search_tensors(q, 10) for q in qs]
    torch.cuda.synchronize()
    errors = []

    def worker(i, own_stream):
        try:
            s = torch.cuda.Stream() if own_stream else torch.cuda.current_stream()
            with torch.cuda.stream(s):
                for _ in range(25):
                    d, r = ix.search_tensors(qs[i], 10)
                    s.synchronize()
                    if not (torch.equal(r, ref[i][1]) and torch.equal(d, ref[i][0])):
                        errors.append(f"thread {i} own_stream={own_stream}: wrong result")
                        return
        except Exception as e:  # noqa: BLE001
            errors.append(f"thread {i}: {e!r}")

    for own_stream in (False, True):
        threads = [threading.Thread(target=worker, args=(i, own_stream)) for i in range(4)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    assert not errors, errors
    ix.close()
