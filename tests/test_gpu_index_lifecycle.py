"""Lifecycle properties of the device index that round 2 changed: the bf16 shadow is derived lazily (memory), writes on one
stream are ordered against searches on others on the device, loaded rows are not trusted to be unit vectors."""

import json
import os

import numpy as np
import pytest

from oracle import knn_oracle as o

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch

    assert torch.cuda.is_available()
    from codd_query_engine_amd.knn_index import DeviceKnnIndex

    return torch, DeviceKnnIndex


def test_bf16_shadow_is_built_only_when_a_search_needs_it(env):
    torch, Index = env
    n, d, k = 60_000, 768, 10
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((n, d), generator=g, device="cuda")
    ix = Index(d)
    ix.upsert_device(0, x)
    rows_bytes = n * d * 4
    q = torch.randn((256, d), generator=g, device="cuda")
    d8, r8 = ix.search_tensors(q, k)                      # int8 filter: no bf16 shadow
    assert ix.stat("shadow8_builds") == 1 and ix.stat("shadow16_builds") == 0
    before = ix.stat("device_bytes")                       # rows + int8 shadow + workspaces: no 2-byte copy of the corpus yet
    q_big = torch.randn((300, d), generator=g, device="cuda")
    ix.set_option("shadow8_max_batch", 64)                 # 300 queries: two passes through the bf16 filter
    d16, r16 = ix.search_tensors(q_big, k)
    assert ix.stat("shadow16_builds") == 1
    assert ix.stat("device_bytes") - before >= rows_bytes // 2   # ... now there is one (n x d x 2 bytes, whole tiles, head room)
    ix.set_option("filter", 0)
    d_e, r_e = ix.search_tensors(q_big, k)
    ix.set_option("filter", 1)
    assert torch.equal(r16, r_e) and torch.equal(d16, d_e)
    # rows written later reach both shadows incrementally (only the dirty range is converted again)
    upd = torch.randn((100, d), generator=g, device="cuda")
    ix.upsert_device(5_000, upd)
    d2, r2 = ix.search_tensors(upd[:80].contiguous() * 3.0, 1)          # bf16 path (80 > 64)
    assert r2[:, 0].tolist() == list(range(5_000, 5_080)) and ix.stat("shadow16_builds") == 2
    ix.set_option("shadow8_max_batch", 256)
    d3, r3 = ix.search_tensors(upd[:80].contiguous(), 1)                # int8 path
    assert r3[:, 0].tolist() == list(range(5_000, 5_080)) and ix.stat("shadow8_builds") == 2
    ix.close()


def test_upsert_on_one_stream_is_seen_by_searches_on_another(env):
    """codd_knn_upsert_device is asynchronous on the caller's stream; a search issued right behind it on ANOTHER stream
    must wait for the write on the device (and rebuild its shadow from the new rows), with no host synchronisation."""
    torch, Index = env
    n, d, k = 400_000, 256, 5
    g = torch.Generator(device="cuda").manual_seed(2)
    ix = Index(d)
    ix.upsert_device(0, torch.randn((n, d), generator=g, device="cuda"))
    q = torch.randn((64, d), generator=g, device="cuda")
    ix.search_tensors(q, k)
    torch.cuda.synchronize()
    s_write, s_read = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(3):
        first = 1_000 + rep * 10_000
        q = torch.randn((64, d), generator=g, device="cuda")   # (new queries every time: the rows of an earlier round must not tie)
        fresh = q * 2.0                                         # rows identical (after normalisation) to the queries
        torch.cuda.synchronize()
        with torch.cuda.stream(s_write):
            filler = torch.randn((200_000, d), generator=g, device="cuda")   # keeps the writing stream busy first
            ix.upsert_device(100_000, filler)
            ix.upsert_device(first, fresh.contiguous())
        with torch.cuda.stream(s_read):
            dist, rows = ix.search_tensors(q, k)
        s_read.synchronize()
        assert rows[:, 0].tolist() == list(range(first, first + 64)), rep
        assert float(dist[:, 0].abs().max()) < 1e-6
    torch.cuda.synchronize()
    ix.close()


def test_two_writer_streams_then_a_reader_on_a_third(env):
    """rows_ready is ONE event, re-recorded by every asynchronous write.  Stream A upserts (behind a long filler), stream
    B upserts next, a search on stream C waits for B's record only — so B must have waited for A on the device, or C reads rows
    that A has not finished writing.  copy_rows_f32 (the IVF build's reader) is ordered the same way, and a later writer on
    another stream waits for it."""
    torch, Index = env
    import ctypes
    from codd_query_engine_amd import native

    n, d, k = 300_000, 256, 3
    g = torch.Generator(device="cuda").manual_seed(5)
    ix = Index(d)
    ix.upsert_device(0, torch.randn((n, d), generator=g, device="cuda"))
    ix.search_tensors(torch.randn((64, d), generator=g, device="cuda"), k)
    torch.cuda.synchronize()
    s_a, s_b, s_c = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    lib = native.load()
    for rep in range(3):
        qa = torch.randn((64, d), generator=g, device="cuda")
        qb = torch.randn((64, d), generator=g, device="cuda")
        first_a, first_b = 10_000 + rep * 1_000, 50_000 + rep * 1_000
        torch.cuda.synchronize()
        with torch.cuda.stream(s_a):
            filler = torch.randn((250_000, d), generator=g, device="cuda")     # keeps stream A busy in front of its write
            ix.upsert_device(20_000, filler)
            ix.upsert_device(first_a, (qa * 2.0).contiguous())
        with torch.cuda.stream(s_b):
            ix.upsert_device(first_b, (qb * 3.0).contiguous())               # the LAST writer: its event is what readers wait for
        with torch.cuda.stream(s_c):
            dist_a, rows_a = ix.search_tensors(qa, k)
            dist_b, rows_b = ix.search_tensors(qb, k)
            out = torch.empty((64, d), dtype=torch.float32, device="cuda")
            native.check(lib.codd_knn_copy_rows_f32(ix._h, first_a, 64, out.data_ptr(), ctypes.c_void_p(s_c.cuda_stream)), "copy_rows")
        s_c.synchronize()
        assert rows_a[:, 0].tolist() == list(range(first_a, first_a + 64)), rep
        assert rows_b[:, 0].tolist() == list(range(first_b, first_b + 64)), rep
        assert float(dist_a[:, 0].abs().max()) < 1e-6 and float(dist_b[:, 0].abs().max()) < 1e-6
        want = qa / qa.norm(dim=1, keepdim=True)
        assert float((out - want).abs().max()) < 1e-6, rep                  # the reader saw A's rows, not what was there before
    torch.cuda.synchronize()
    ix.close()


def test_a_bf16_shadow_that_cannot_be_allocated_leaves_results_exact(env):
    """the bf16 shadow is allocated by the first search that leaves the int8 path (+15 GB at 10M x 768).  When that
    hipMalloc fails the search is answered exactly another way — the int8 filter where the index may use it, the exact scan
    otherwise — and the allocation is not retried until rows change."""
    torch, Index = env
    n, d, k = 80_000, 768, 10
    g = torch.Generator(device="cuda").manual_seed(6)
    ix = Index(d)
    ix.upsert_device(0, torch.randn((n, d), generator=g, device="cuda"))
    q = torch.randn((300, d), generator=g, device="cuda")
    ix.set_option("filter", 0)
    d_ref, r_ref = ix.search_tensors(q, k)
    ix.set_option("filter", 1)
    ix.set_option("debug_fail_shadow_alloc", 1)
    ix.set_option("shadow8_max_batch", 64)                 # 300 queries: the bf16 filter's batch -> its shadow is wanted, and "HBM is full"
    scans = ix.stat("scan_launches")
    d1, r1 = ix.search_tensors(q, k)                        # (the int8 filter is not allowed this batch size: the exact scan answers)
    assert torch.equal(r1, r_ref) and torch.equal(d1, d_ref) and ix.stat("scan_launches") > scans
    assert ix.stat("shadow16_alloc_failures") == 1 and ix.stat("shadow16_builds") == 0
    d2, r2 = ix.search_tensors(q, k)                        # not retried: one failure on record
    assert torch.equal(r2, r_ref) and ix.stat("shadow16_alloc_failures") == 1
    ix.set_option("shadow8", 0)                             # an index without the int8 filter: same
    d3, r3 = ix.search_tensors(q[:200].contiguous(), k)
    assert torch.equal(r3, r_ref[:200]) and torch.equal(d3, d_ref[:200])
    ix.set_option("shadow8", 1)
    ix.upsert_device(0, torch.randn((8, d), generator=g, device="cuda"))   # rows changed: the allocation is tried again (and fails again)
    ix.search_tensors(q, k)
    assert ix.stat("shadow16_alloc_failures") == 2
    ix.set_option("debug_fail_shadow_alloc", 0)            # memory is back: the shadow is built and used
    d4, r4 = ix.search_tensors(q, k)
    assert ix.stat("shadow16_builds") == 1
    ix.set_option("filter", 0)
    d5, r5 = ix.search_tensors(q, k)
    assert torch.equal(r4, r5) and torch.equal(d4, d5)
    ix.close()


def test_loaded_rows_that_are_not_unit_vectors_switch_the_filters_off(env, tmp_path):
    torch, Index = env
    from codd_query_engine_amd import KnnClient

    n, d, k = 40_000, 128, 10
    rng = np.random.default_rng(3)
    raw = rng.standard_normal((n, d)).astype(np.float32)
    raw[::7] *= 3.0                                        # stored as they are: norms far from 1
    q = rng.standard_normal((32, d)).astype(np.float32)
    a = Index(d)
    a.upsert(np.arange(n, dtype=np.int64), raw, normalize=False)
    assert a.stat("all_normalized") == 0
    d_a, r_a = a.search(q, k)
    d_ref, i_ref = o.search(raw, "f32", o.normalize_rows(q), k)     # oracle on the rows as stored
    assert np.array_equal(r_a, i_ref) and np.array_equal(d_a, d_ref)
    b = Index(d)
    b.load_rows(a.read_rows())                              # a reader trusts nothing: the norm check finds them
    assert b.stat("all_normalized") == 0
    d_b, r_b = b.search(q, k)
    assert b.stat("filter_passes") == 0
    assert np.array_equal(r_b, i_ref) and np.array_equal(d_b, d_ref)
    c = Index(d)
    c.load_rows(o.normalize_rows(raw))                      # unit rows: filters stay on
    assert c.stat("all_normalized") == 1
    for ix in (a, b, c):
        ix.close()
    # the manifest carries the flag too: a generation written as "not all normalised" loads with the filters off
    client = KnnClient(path=str(tmp_path), device="cuda:0")
    col = client.get_or_create_collection("m", metadata={"hnsw:space": "cosine"})
    col.upsert(documents=["http latency", "free memory"], metadatas=[{"a": "1"}, {"a": "2"}], ids=["x", "y"])
    client.persist()
    gdir = os.path.join(str(tmp_path), "m")
    gen = open(os.path.join(gdir, "CURRENT")).read().strip()
    mpath = os.path.join(gdir, gen, "manifest.json")
    manifest = json.load(open(mpath))
    assert manifest["all_normalized"] is True
    manifest["all_normalized"] = False
    json.dump(manifest, open(mpath, "w"))
    reader = KnnClient(path=str(tmp_path), device="cuda:0")
    assert reader.get_or_create_collection("m")._engine.stat("all_normalized") == 0
    assert reader.get_or_create_collection("m").query(query_texts=["latency"], n_results=1)["ids"] == [["x"]]
