"""BASELINE configs[3] and [4] on the workload one GPU of eight sees, through the code path the 8-GPU job runs
(the N > 1 hardware run is the driver's; the cross-rank exchange is covered by tests/test_sharded_gloo.py):

  [3]  10M x 768 bf16 row-sharded over 8 GPUs  ->  1.25M x 768 bf16 per rank, ShardedSearcher(always_gather=True) over
       single-rank RCCL (the all_gather and the shard merge run), B = 1 and B = 256
  [4]  100M x 1024 fp16 coarse-IVF over 8 GPUs ->  12.5M x 1024 fp16 per rank, nlist 2048

Sizes are the real ones, so the checks are size-independent properties (planted neighbours, filter == exact scan,
two half-shards merged == the whole shard, exhaustive probe == flat search, recall) plus, for IVF, the CPU oracle on a
50k-row corpus."""

import numpy as np
import pytest

from oracle import knn_oracle as o

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch

    assert torch.cuda.is_available()
    from codd_query_engine_amd import ivf
    from codd_query_engine_amd.knn_index import DeviceKnnIndex, merge_shards

    return torch, DeviceKnnIndex, ivf, merge_shards


@pytest.fixture(scope="module")
def rccl_single_rank():
    """A one-rank "nccl" (= RCCL) process group: the collective of the sharded path really runs."""
    import os

    import torch
    import torch.distributed as dist

    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29547")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    yield dist
    if created:
        dist.destroy_process_group()


def fill_randn(torch, ix, n, d, seed, chunk=250_000):
    ix.reserve(n)
    for c0 in range(0, n, chunk):
        m = min(chunk, n - c0)
        g = torch.Generator(device="cuda").manual_seed(seed + c0 // chunk)
        ix.upsert_device(c0, torch.randn((m, d), generator=g, device="cuda"))
    torch.cuda.synchronize()


def test_config3_bf16_shard_through_the_sharded_searcher(env, rccl_single_rank):
    torch, Index, _, merge_shards = env
    from codd_query_engine_amd.sharded import ShardedSearcher

    n, d, k, row_base = 1_250_000, 768, 10, 3 * 1_250_000   # rank 3 of 8
    ix = Index(d, "bf16")
    fill_randn(torch, ix, n, d, seed=300)
    g = torch.Generator(device="cuda").manual_seed(9)
    q = torch.randn((256, d), generator=g, device="cuda")
    # plant 10 near-duplicates of queries 0..3 at known local rows
    planted = {}
    for b in range(4):
        rows = [(b * 123_457 + j * 9_973 + 11) % n for j in range(k)]
        for j, r in enumerate(rows):
            noise = torch.randn(d, generator=g, device="cuda")
            ix.upsert_device(r, (q[b] + 0.02 * (j + 1) * noise * q[b].norm() / noise.norm())[None, :].contiguous())
        planted[b] = rows
    searcher = ShardedSearcher(ix, row_base=row_base, always_gather=True)
    # B = 256: the planted rows come back first, in order, with GLOBAL row ids
    dist256, rows256 = searcher.search(q, k)
    for b, rows in planted.items():
        assert rows256[b].tolist() == [row_base + r for r in rows]
    assert bool((dist256[:, 1:] >= dist256[:, :-1]).all())
    # the filter path == the exact scan of the same shard, every query, ids and distances
    d_f, r_f = ix.search_tensors(q, k)
    assert ix.stat("filter_passes") >= 1
    ix.set_option("filter", 0)
    d_e, r_e = ix.search_tensors(q, k)
    ix.set_option("filter", 1)
    assert torch.equal(r_f, r_e) and torch.equal(d_f, d_e)
    assert torch.equal(rows256, r_e + row_base) and torch.equal(dist256, d_e)
    # B = 1 (the latency point of the config)
    d1, r1 = searcher.search(q[:1].contiguous(), k)
    assert r1[0].tolist() == [row_base + r for r in planted[0]] and torch.equal(d1[0], d_e[0])
    # two half-shards, each searched for keys with its own row base, merged by codd_knn_merge_shards == the whole shard
    half = n // 2
    stored = ix.read_rows(0, n)
    a, b = Index(d, "bf16"), Index(d, "bf16")
    a.load_rows(stored[:half]); b.load_rows(stored[half:])
    del stored
    ka = a.search_keys(q, k, row_base)
    kb = b.search_keys(q, k, row_base + half)
    _, d_m, r_m = merge_shards(torch.cat([ka, kb], dim=0), 2, k)
    assert torch.equal(r_m, r_e + row_base) and torch.equal(d_m, d_e)
    for x in (ix, a, b):
        x.close()


def clustered(torch, n, d, centres, seed, noise=0.3, chunk=500_000):
    gc = torch.Generator(device="cuda").manual_seed(7)
    c = torch.nn.functional.normalize(torch.randn((centres, d), generator=gc, device="cuda"), dim=1)
    for c0 in range(0, n, chunk):
        m = min(chunk, n - c0)
        g = torch.Generator(device="cuda").manual_seed(seed + c0 // chunk)
        which = torch.randint(0, centres, (m,), generator=g, device="cuda")
        yield c0, (c[which] + noise * torch.randn((m, d), generator=g, device="cuda") / d ** 0.5).contiguous()


def test_config4_fp16_ivf_shard(env):
    torch, Index, ivf, _ = env
    n, d, k, nlist = 12_500_000, 1024, 10, 2048
    ix = Index(d, "f16")
    ix.reserve(n)
    for c0, x in clustered(torch, n, d, 4096, seed=40):
        ix.upsert_device(c0, x)
    torch.cuda.synchronize()
    stats = ivf.build_ivf(ix, nlist, iters=4)
    assert stats["rows"] == n and stats["max_list"] < n // 8
    q = next(clustered(torch, 64, d, 4096, seed=77))[1]
    d_flat, r_flat = ix.search_tensors(q, k)
    # recall@10 at nprobe 32 against the exact answer of the same shard; the hits' distances are the exact ones
    d_ivf, r_ivf = ivf.search_ivf(ix, q, k, 32)
    recall = (r_ivf.unsqueeze(2) == r_flat.unsqueeze(1)).any(dim=2).float().mean().item()
    assert recall >= 0.95, recall
    same = r_ivf == r_flat
    assert torch.equal(d_ivf[same], d_flat[same])
    # probing EVERY list == the flat search, bit for bit, at full shard size: one call probes at most 128 lists, so the
    # identity is checked on a second layout of the same rows with 128 lists (8 queries: an exhaustive IVF pass reads
    # the shard once per query)
    ivf.build_ivf(ix, 128, iters=2)
    d_all, r_all = ivf.search_ivf(ix, q[:8].contiguous(), k, 128)
    assert torch.equal(r_all, r_flat[:8]) and torch.equal(d_all, d_flat[:8])
    ix.close()


def test_ivf_against_the_cpu_oracle(env):
    """IVF results checked against the ORACLE (not against this engine's own flat search): exhaustive probing must equal
    the oracle's exact answer bit for bit; a partial probe returns oracle-exact distances for every row it returns, and
    its recall against the oracle's ids is high on clustered rows."""
    torch, Index, ivf, _ = env
    n, d, k, nlist = 50_000, 256, 10, 64
    rows = torch.cat([x for _, x in clustered(torch, n, d, 128, seed=5, chunk=50_000)])
    q = next(clustered(torch, 40, d, 128, seed=6))[1]
    raw, q_raw = rows.cpu().numpy(), q.cpu().numpy()
    for dtype in ("f32", "f16"):
        ix = Index(d, dtype)
        ix.upsert_device(0, rows)
        ivf.build_ivf(ix, nlist, iters=4)
        stored = o.to_storage(o.normalize_rows(raw), dtype)
        assert np.array_equal(ix.read_rows(), stored)
        d_ref, i_ref = o.search(stored, dtype, o.normalize_rows(q_raw), k)
        d_all, r_all = ivf.search_ivf(ix, q, k, nprobe=nlist)
        assert np.array_equal(r_all.cpu().numpy(), i_ref) and np.array_equal(d_all.cpu().numpy(), d_ref)
        d_p, r_p = ivf.search_ivf(ix, q, k, nprobe=8)
        r_p, d_p = r_p.cpu().numpy(), d_p.cpu().numpy()
        hits = 0
        for b in range(q.shape[0]):
            ref = {int(r): float(dd) for r, dd in zip(i_ref[b], d_ref[b])}
            for r, dd in zip(r_p[b], d_p[b]):
                if int(r) in ref:
                    hits += 1
                    assert float(dd) == ref[int(r)]          # oracle-exact distance for every true neighbour returned
            assert (np.diff(d_p[b]) >= 0).all()
        assert hits / i_ref.size >= 0.9
        ix.close()


def test_ivf_install_rejects_bad_tables(env):
    """codd_knn_ivf_install validates what it is handed (offsets on the host, the permutation on the device) and leaves no
    half-built layout behind."""
    import ctypes

    torch, Index, ivf, _ = env
    from codd_query_engine_amd import native

    lib = native.load()
    n, d, nlist = 4096, 64, 8
    ix = Index(d)
    ix.upsert_device(0, torch.randn((n, d), device="cuda"))
    cent = torch.randn((nlist, d), device="cuda")
    perm = torch.arange(n, dtype=torch.int64, device="cuda")
    good = torch.arange(0, n + 1, n // nlist, dtype=torch.int64, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def install(p, off):
        return lib.codd_knn_ivf_install(ix._h, cent.data_ptr(), nlist, p.data_ptr(), off.data_ptr(), st)

    bad_end = good.clone(); bad_end[-1] = n - 1
    decreasing = good.clone(); decreasing[3] = decreasing[2] - 1
    bad_perm = perm.clone(); bad_perm[17] = n
    neg_perm = perm.clone(); neg_perm[5] = -1
    for p, off in ((perm, bad_end), (perm, decreasing), (bad_perm, good), (neg_perm, good)):
        assert install(p, off) == -22      # CODD_KNN_EINVAL
        q = torch.randn((3, d), device="cuda")
        with pytest.raises(native.NativeLibraryError):
            ivf.search_ivf(ix, q, 5, 2)          # nothing was installed
    assert install(perm, good) == 0
    d_i, r_i = ivf.search_ivf(ix, torch.randn((3, d), device="cuda"), 5, nlist)
    assert int(r_i.min()) >= 0
    ix.close()
