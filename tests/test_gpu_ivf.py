"""Coarse-IVF + exact scores (BASELINE config 5).  Approximate by design: judged by
recall@10 against this engine's exact search; exhaustive probing must reproduce it bit for bit."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch

    assert torch.cuda.is_available()
    from codd_query_engine_amd import ivf
    from codd_query_engine_amd.knn_index import DeviceKnnIndex

    return torch, DeviceKnnIndex, ivf


def clustered(torch, n, d, centres, seed=7, noise=0.5, centre_seed=7):
    """Clustered corpus in the spirit of SURVEY §8d row 5: rows = unit centre + noise of norm ~`noise`
    (cosine to the own centre ~0.9), normalised at ingest.  Queries are drawn the same way."""
    gc = torch.Generator(device="cuda").manual_seed(centre_seed)
    c = torch.nn.functional.normalize(torch.randn((centres, d), generator=gc, device="cuda"), dim=1)
    g = torch.Generator(device="cuda").manual_seed(seed + 1000)
    which = torch.randint(0, centres, (n,), generator=g, device="cuda")
    return c[which] + noise * torch.randn((n, d), generator=g, device="cuda") / (d ** 0.5)


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_exhaustive_probe_equals_flat_search(env, dtype):
    torch, Index, ivf = env
    n, d, nlist = 20_000, 128, 32
    x = clustered(torch, n, d, 64)
    ix = Index(d, dtype)
    ix.upsert_device(0, x.contiguous())
    stats = ivf.build_ivf(ix, nlist, iters=4)
    assert stats["rows"] == n and stats["min_list"] >= 0 and stats["max_list"] <= n
    q = clustered(torch, 9, d, 64, seed=11)
    for k in (1, 10, 100):
        d_flat, r_flat = ix.search_tensors(q, k)
        d_ivf, r_ivf = ivf.search_ivf(ix, q, k, nprobe=nlist)
        assert torch.equal(r_ivf, r_flat) and torch.equal(d_ivf, d_flat)
    ix.close()


def test_recall_on_clustered_data_and_staleness(env):
    torch, Index, ivf = env
    n, d, nlist, k = 200_000, 128, 256, 10
    x = clustered(torch, n, d, 512)
    ix = Index(d, "f16")
    ix.upsert_device(0, x.contiguous())
    ivf.build_ivf(ix, nlist, iters=6)
    q = clustered(torch, 64, d, 512, seed=13)
    _, truth = ix.search_tensors(q, k)
    recalls = {}
    for nprobe in (1, 8, 32):
        _, got = ivf.search_ivf(ix, q, k, nprobe)
        hit = (got.unsqueeze(2) == truth.unsqueeze(1)).any(dim=2).float().mean().item()
        recalls[nprobe] = hit
    assert recalls[1] < recalls[8] <= recalls[32] + 1e-9
    assert recalls[32] >= 0.95, recalls
    # distances reported for the hits are the exact canonical ones
    d_ivf, r_ivf = ivf.search_ivf(ix, q, k, 32)
    d_flat, r_flat = ix.search_tensors(q, k)
    same = r_ivf == r_flat
    assert torch.equal(d_ivf[same], d_flat[same])
    # a later upsert makes the layout stale: the search refuses instead of answering from old rows
    ix.upsert_device(5, x[:1].contiguous())
    from codd_query_engine_amd.native import NativeLibraryError

    with pytest.raises(NativeLibraryError):
        ivf.search_ivf(ix, q, k, 8)
    ix.close()


def test_two_ivf_shards_merge_to_the_whole_answer(env):
    """configs[4]'s composition on one GPU: two shards, each with its own IVF, probed exhaustively, keys carrying global
    rows, merged by codd_knn_merge_keys == the flat search over all rows (the cross-rank gather is covered by the gloo tests)."""
    torch, Index, ivf = env
    from codd_query_engine_amd.sharded import ShardedSearcher

    n, d, k, cut = 30_000, 128, 10, 13_000
    x = clustered(torch, n, d, 64).contiguous()
    whole = Index(d, "f16")
    whole.upsert_device(0, x)
    q = clustered(torch, 17, d, 64, seed=5)
    d_ref, r_ref = whole.search_tensors(q, k)
    shards = []
    for lo, hi in ((0, cut), (cut, n)):
        s = Index(d, "f16")
        s.upsert_device(0, x[lo:hi].contiguous())
        ivf.build_ivf(s, 16, iters=3)
        shards.append((lo, s))
    keys = torch.cat([ivf.IvfShardEngine(s, nprobe=16).search_keys(q, k, lo) for lo, s in shards], dim=1)
    _, d_got, r_got = whole.merge_keys(keys, k)
    assert torch.equal(r_got, r_ref) and torch.equal(d_got, d_ref)
    # and through ShardedSearcher (one rank: no collective), with a partial probe: global rows, sorted, mostly the true ones
    lo, s = shards[1]
    d1, r1 = ShardedSearcher(ivf.IvfShardEngine(s, nprobe=4), row_base=lo).search(q, k)
    assert int(r1.min()) >= lo and bool((d1[:, 1:] >= d1[:, :-1]).all())
    _, r_shard = s.search_tensors(q, k)
    recall = (r1.unsqueeze(2) == (r_shard + lo).unsqueeze(1)).any(dim=2).float().mean().item()
    assert recall >= 0.9
    for ix in (whole, shards[0][1], shards[1][1]):
        ix.close()


@pytest.mark.parametrize(
    "n,d,nlist,B,nprobe,k,dtype",
    [
        (60_000, 128, 64, 256, 8, 10, "f16"),     # 2,048 pairs over 64 lists: ~32 queries per list, 8 work items each
        (60_000, 128, 64, 40, 64, 10, "f32"),     # exhaustive probe through the shared scan: must equal the flat search as well
        (50_000, 768, 128, 128, 16, 100, "bf16"),  # k above 64: two top-k slots per lane
        (30_000, 1024, 32, 256, 4, 10, "f16"),    # config 5's row width
        (20_000, 100, 300, 70, 16, 10, "f32"),    # more lists than any query group touches; ragged row width
    ],
)
def test_list_sharing_scan_returns_the_per_pair_scan_s_bits(env, n, d, nlist, B, nprobe, k, dtype):
    """From 1,024 (query, list) pairs on, the pairs are grouped by list on the device and every probed list is read once per
    group of its queries.  Same candidates, same canonical scores, same merge: the answer must be the per-pair scan's, bit for bit."""
    torch, Index, ivf = env
    x = clustered(torch, n, d, 2 * nlist)
    ix = Index(d, dtype)
    ix.upsert_device(0, x.contiguous())
    ivf.build_ivf(ix, nlist, iters=3)
    q = clustered(torch, B, d, 2 * nlist, seed=21)
    q[B // 2 :] = q[0]                      # half of the batch probes the very same lists: one list, >100 pairs
    q[1] = 0.0                              # and a zero query
    ix.set_option("ivf_share", 0)
    d0, r0 = ivf.search_ivf(ix, q, k, nprobe)
    ix.set_option("ivf_share", 1)
    d1, r1 = ivf.search_ivf(ix, q, k, nprobe)
    assert torch.equal(r1, r0) and torch.equal(d1, d0)
    d2, r2 = ivf.search_ivf(ix, q, k, nprobe)   # the grouping scratch is reused: a second call must not see the first one's counters
    assert torch.equal(r2, r0) and torch.equal(d2, d0)
    if nprobe == nlist:
        df, rf = ix.search_tensors(q, k)
        assert torch.equal(r1, rf) and torch.equal(d1, df)
    ix.close()
