#!/usr/bin/env python3
"""
bench.py — headline benchmark of the search_relevant_metrics hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json metric "queries/sec + p50 latency, top-10 over 10M x 768 corpus at
1/2/4/8 GPUs", quoted on configs[2]): a synthetic 10M x 768 fp32 corpus resident in HBM,
row-sharded over the N ranks; one STEP = one batch of 256 queries answered with their exact
top-10 (shard-local search, one RCCL all_gather of the packed partials, integer merge).
`value` = whole-job queries/s with queries and corpus already in HBM; results stay in HBM.
The p50 single-query (B=1) latency, host wall clock including the D2H of the 10 results, is
reported next to it.  Prints ONE JSON line on rank 0.
"""

from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (guide: ~6290 GB/s achievable)
CHUNK_ROWS = 250_000


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--rows", type=int, default=10_000_000)
    p.add_argument("--dim", type=int, default=768)
    p.add_argument("--batch", type=int, default=256)
    p.add_argument("--k", type=int, default=10)
    p.add_argument("--dtype", default="f32", choices=["f32", "bf16", "f16"])
    p.add_argument("--latency-iters", type=int, default=30)
    p.add_argument("--cpu-sample-rows", type=int, default=4_000_000)
    p.add_argument("--cpu-sample-queries", type=int, default=256)
    p.add_argument("--cpu-seconds", type=float, default=10.0)
    p.add_argument("--hnsw-build-seconds", type=float, default=12.0, help="target build time of the HNSW comparator's sample")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--validate-before", action="store_true", help="run the correctness gate (64 exact scans) in front of the warm-up instead of behind the timed steps")
    p.add_argument("--set", action="append", default=[], metavar="KEY=VALUE",
                   help="engine option for A/B runs (codd_knn_set_option), e.g. --set i8v2=2; recorded in the line")
    p.add_argument("--corpus", default="isotropic", choices=["isotropic", "clustered"],
                   help="isotropic: seeded randn rows (the headline workload); clustered: unit centres + noise, queries drawn the "
                        "same way (what embedding corpora look like: dense neighbourhoods near the top of the score distribution)")
    p.add_argument("--centres", type=int, default=4096)
    p.add_argument("--noise", type=float, default=0.5, help="clustered: rows = centre + noise * randn / sqrt(dim)")
    p.add_argument("--pipeline", type=int, default=0,
                   help="batches in flight on alternating HIP streams in the timed loop (0 = 1 on one GPU, 2 when sharded)")
    p.add_argument("--force-dist", action="store_true",
                   help="initialise RCCL and take the sharded code path even with one rank (rehearsal on a 1-GPU box)")
    return p.parse_args()


def planted_row(b: int, j: int, n_total: int) -> int:
    return (b * 1_234_567 + j * 99_991 + 17) % n_total


def cluster_centres(torch, args, device):
    g = torch.Generator(device=device).manual_seed(7)
    return torch.nn.functional.normalize(torch.randn((args.centres, args.dim), generator=g, device=device), dim=1)


def draw(torch, n, dim, g, device, centres=None, noise=0.0):
    """n synthetic vectors from generator g: randn, or (clustered) a random centre + noise * randn / sqrt(dim)."""
    x = torch.randn((n, dim), generator=g, device=device, dtype=torch.float32)
    if centres is None:
        return x
    which = torch.randint(0, centres.shape[0], (n,), generator=g, device=device)
    return centres[which] + (noise / dim ** 0.5) * x


def build_shard(ix, torch, lo, hi, n_total, dim, queries, n_planted_q, k, device, centres=None, spread=0.0):
    """Fill rows [lo,hi) of the GLOBAL corpus: chunk c (rows [c*CH,(c+1)*CH)) is drawn with seed
    1234+c, so the corpus is identical for every rank count; then the planted near-duplicates."""
    ix.reserve(max(hi - lo, 1))
    c0, c1 = lo // CHUNK_ROWS, (max(hi, 1) - 1) // CHUNK_ROWS
    for c in range(c0, c1 + 1):
        a, b = c * CHUNK_ROWS, min(n_total, (c + 1) * CHUNK_ROWS)
        if b <= lo or a >= hi:
            continue
        g = torch.Generator(device=device).manual_seed(1234 + c)
        x = draw(torch, b - a, dim, g, device, centres, spread)
        s, e = max(a, lo), min(b, hi)
        ix.upsert_device(s - lo, x[s - a : e - a].contiguous(), normalize=True)
        torch.cuda.synchronize()
        del x
    gp = torch.Generator(device=device).manual_seed(99)
    for b in range(n_planted_q):
        for j in range(k):
            noise = torch.randn(dim, generator=gp, device=device)  # drawn on every rank: same stream
            row = planted_row(b, j, n_total)
            if lo <= row < hi:
                vec = queries[b] + (0.02 * (j + 1)) * noise * queries[b].norm() / noise.norm()
                ix.upsert_device(row - lo, vec[None, :].contiguous(), normalize=True)
    torch.cuda.synchronize()


def cpu_baseline(ix, args, queries_cpu):
    """The oracle's all-core fp32 scan (port) on a bounded sample of the same corpus: a short probe
    sizes the sample so that the timed run is ~10 s of CPU work on whatever host this is."""
    import numpy as np

    from oracle import knn_oracle as o

    nq = min(args.cpu_sample_queries, queries_cpu.shape[0])
    qn = o.normalize_rows(np.ascontiguousarray(queries_cpu[:nq]))
    probe_n = min(100_000, ix.count())
    rows = ix.read_rows(0, probe_n)
    rows = rows if ix.dtype == "f32" else o.widen(rows, ix.dtype)
    o.search_fast_f32(rows[:2000], qn, args.k)  # warm the thread pool
    t0 = time.perf_counter()
    o.search_fast_f32(rows, qn, args.k)
    rate = nq * probe_n / max(time.perf_counter() - t0, 1e-6)  # dot products / s
    n = int(min(ix.count(), args.cpu_sample_rows, max(probe_n, args.cpu_seconds * rate / nq)))
    if n > probe_n:
        rows = ix.read_rows(0, n)
        rows = rows if ix.dtype == "f32" else o.widen(rows, ix.dtype)
    t0 = time.perf_counter()
    o.search_fast_f32(rows, qn, args.k)
    dt = time.perf_counter() - t0
    qps_full = (nq * n / dt) / args.rows  # same work per row: scale the sample to the full corpus
    # SURVEY.md §8(d) comparator 3: the same sample as one fp32 GEMM + topk on torch's CPU backend (all threads).  Not the
    # canonical arithmetic (blocked summation order), so it is a speed comparator only, never a checker.
    gemm = None
    try:
        import torch

        rt, qt = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.float32)), torch.from_numpy(qn)
        (qt @ rt[:4096].T).topk(args.k, dim=1)
        t1 = time.perf_counter()
        best = None
        for c0 in range(0, n, 1_000_000):  # chunked: the [nq, n] score block never exists in full
            sc, ix_ = (qt @ rt[c0 : c0 + 1_000_000].T).topk(min(args.k, rt[c0 : c0 + 1_000_000].shape[0]), dim=1)
            cand = (sc, ix_ + c0)
            if best is not None:
                sc2, sel = torch.cat([best[0], cand[0]], 1).topk(args.k, dim=1)
                cand = (sc2, torch.cat([best[1], cand[1]], 1).gather(1, sel))
            best = cand
        dt_g = time.perf_counter() - t1
        gemm = {"value": (nq * n / dt_g) / args.rows, "unit": "queries/s", "threads": torch.get_num_threads(),
                "what": "torch CPU: chunked Q @ C^T + topk on the same sample, scaled the same way"}
    except Exception as e:  # noqa: BLE001
        gemm = {"value": None, "what": f"failed: {e}"}
    # SURVEY.md §8(d) comparator 3b: scikit-learn's brute-force cosine neighbours on <= 1M rows of the same sample
    skl = None
    try:
        from sklearn.neighbors import NearestNeighbors

        n_s = int(min(n, 1_000_000))
        nn = NearestNeighbors(n_neighbors=args.k, metric="cosine", algorithm="brute", n_jobs=-1).fit(rows[:n_s])
        t2 = time.perf_counter()
        nn.kneighbors(qn, return_distance=True)
        dt_s = time.perf_counter() - t2
        skl = {"value": (nq * n_s / dt_s) / args.rows, "unit": "queries/s", "rows": n_s,
               "what": "sklearn NearestNeighbors(metric='cosine', algorithm='brute') on the first rows of the same sample, scaled the same way"}
    except Exception as e:  # noqa: BLE001
        skl = {"value": None, "what": f"failed: {e}"}
    # the like-for-like ALGORITHM comparator — a from-scratch HNSW with the reference's parameters
    # (M = 16, construction_ef = 200, search_ef = 100; store.py:63-68) on a sample sized so that its build takes ~10 s,
    # with its recall@10 against the exact answer on that sample.  ChromaDB's index is approximate; this engine is exact.
    hn = None
    try:
        from oracle import hnsw_cpu

        probe = min(20_000, n)
        t3 = time.perf_counter()
        hnsw_cpu.HnswIndex(rows[:probe]).close()
        per_row = (time.perf_counter() - t3) / probe
        n_h = int(min(n, 1_000_000, max(probe, args.hnsw_build_seconds / per_row / 1.3)))  # (1.3: cost per row grows ~log n)
        t3 = time.perf_counter()
        hx = hnsw_cpu.HnswIndex(rows[:n_h])
        build_s = time.perf_counter() - t3
        hx.search(qn[:8], args.k)
        _, exact_ids = o.search_fast_f32(rows[:n_h], qn, args.k)
        # A throughput figure only means something next to its recall: on the isotropic headline corpus the reference's
        # search_ef = 100 finds one true neighbour in ten.  `value` is therefore the rate AT MATCHED QUALITY — search_ef doubled
        # until recall@10 >= 0.95 (capped at the sample size = exhaustive) — and the reference-parameter point is kept as an
        # observation of recall, not as a speed.
        t3 = time.perf_counter()
        _, ids = hx.search(qn, args.k)
        dt_ref = time.perf_counter() - t3
        at_ref = {"search_ef": hnsw_cpu.REFERENCE_PARAMS["search_ef"], "queries_per_s": nq / dt_ref, "recall_at_10": hnsw_cpu.recall_at_k(ids, exact_ids)}
        ef, rec, dt_h = at_ref["search_ef"], at_ref["recall_at_10"], dt_ref
        t_budget = time.perf_counter()
        while rec < 0.95 and ef < n_h and time.perf_counter() - t_budget < 20.0:
            ef = min(2 * ef, n_h)
            t3 = time.perf_counter()
            _, ids = hx.search(qn, args.k, search_ef=ef)
            dt_h = time.perf_counter() - t3
            rec = hnsw_cpu.recall_at_k(ids, exact_ids)
        hn = {"kind": "hnsw-restatement (unpinned: a from-scratch restatement of the published algorithm, not hnswlib)",
              "value": nq / dt_h if rec >= 0.95 else None, "unit": "queries/s", "recall_at_10": rec, "search_ef": ef,
              "matched_quality": rec >= 0.95, "at_reference_search_ef": at_ref,
              "cores": o.num_threads(), "rows": n_h, "build_s": build_s, "params": hnsw_cpu.REFERENCE_PARAMS,
              "what": f"oracle/hnsw_cpu.c (M 16, construction_ef 200 as the reference, store.py:63-68) over the first {n_h} rows of the same corpus, {nq} queries; "
                      "`value` = queries/s at the smallest doubled search_ef whose recall@10 against the exact answer on the same rows reaches 0.95 "
                      "(null when 20 s of doubling did not get there); NOT scaled to the full corpus (graph search cost grows ~log N)"}
        hx.close()
    except Exception as e:  # noqa: BLE001
        hn = {"kind": "hnsw-restatement", "value": None, "what": f"failed: {e}"}
    return {
        "also_torch_cpu_gemm_topk": gemm,
        "also_sklearn_brute": skl,
        "hnsw": hn,
        "value": qps_full,
        "unit": "queries/s",
        "cores": o.num_threads(),
        "kind": "port",
        "sample": f"{nq} queries x first {n} rows of the same corpus in {dt:.2f} s, scaled to {args.rows} rows "
                  "(oracle/knn_oracle.c oracle_search_fast_f32, OpenMP; ChromaDB itself is not installable offline)",
    }


class _Synthetic:
    """Row-indexed host bookkeeping (ids / metadata / documents) of the benchmark corpus, produced on demand: 10M Python
    strings and dicts would take minutes and gigabytes to build and are not what is being timed."""

    def __init__(self, n, make):
        self._n, self._make = n, make

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        return self._make(i)


def dropin_call(ix, args, torch):
    """VERDICT r2 missing #3: the latency of the call the reference actually exposes — text in, result dicts out —
    on the resident index: MetricsSearchClient.search_relevant_metrics(text, limit=5) (reference
    codd_lib/codd_lib/client/metrics_promql_client.py:71-107 -> store.py:266-341 -> collection.query) = sanitise, embed on the
    host (HashingEmbeddingFunction at the corpus width), H2D, codd_knn_search, D2H, Chroma-shaped lists, result dicts, 11-key
    projection.  The façade's Collection stands on the SAME engine as the headline number; its ids / metadata are synthesised
    per hit (see _Synthetic).  Outside `value`, like p50_latency_ms_batch1."""
    from codd_query_engine_amd import KnnClient, MetricsSearchClient, MetricsSemanticMetadataStore
    from codd_query_engine_amd.embedding import HashingEmbeddingFunction

    n = ix.count()
    client = KnnClient(device=str(ix.device), embedding_function=HashingEmbeddingFunction(args.dim))
    store = MetricsSemanticMetadataStore(client, collection_name="bench")
    col = store.collection
    col._engine = ix
    col._ids = _Synthetic(n, lambda i: f"prod:bench#metric_{i}")
    col._documents = _Synthetic(n, lambda i: f"synthetic metric {i}")
    col._metadatas = _Synthetic(n, lambda i: {"namespace": "prod:bench", "metric_name": f"metric_{i}", "type": "gauge", "description": f"synthetic metric {i}",
                                              "unit": "seconds", "category": "application", "subcategory": "http", "category_description": "",
                                              "golden_signal_type": "latency", "golden_signal_description": "", "meter_type": "gauge",
                                              "meter_type_description": ""})
    search = MetricsSearchClient(store)
    words = ["http", "request", "latency", "error", "rate", "cpu", "memory", "usage", "disk", "network", "queue", "depth", "database", "query",
             "duration", "seconds", "bytes", "total", "p99", "timeout", "connection", "pool", "gc", "pause", "heap", "cache", "hit", "ratio"]
    rnd = __import__("random").Random(11)
    texts = [" ".join(rnd.choice(words) for _ in range(rnd.randint(3, 12))) for _ in range(256 + args.latency_iters + 3)]
    single, embed1 = [], []
    for i in range(args.latency_iters + 3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = search.search_relevant_metrics(texts[256 + i], limit=5)
        dt = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        col._embed([texts[256 + i]])
        de = (time.perf_counter() - t0) * 1e3
        if i >= 3:
            single.append(dt)
            embed1.append(de)
    assert len(res) == 5 and set(res[0]) >= {"metric_name", "similarity_score"}, res
    batch, embedb = [], []
    for i in range(5 + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        resb = search.search_relevant_metrics_batch(texts[:256], limit=5)
        dt = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        col._embed(texts[:256])
        de = (time.perf_counter() - t0) * 1e3
        if i >= 2:
            batch.append(dt)
            embedb.append(de)
    assert len(resb) == 256 and all(len(r) == 5 for r in resb)
    col._engine = None  # (the index is closed by the caller)
    p50b = statistics.median(batch)
    return {
        "what": "MetricsSearchClient.search_relevant_metrics(text, limit=5) / search_relevant_metrics_batch(256 texts, limit=5) on the same resident "
                f"index ({n} x {args.dim}) through MetricsSemanticMetadataStore -> Collection.query -> codd_knn_search; host wall clock, text in, "
                "11-key result dicts out; ids / metadata synthesised per hit",
        "p50_ms_single": statistics.median(single),
        "of_which_host_embedder_ms_single": statistics.median(embed1),
        "p50_ms_batch256": p50b,
        "of_which_host_embedder_ms_batch256": statistics.median(embedb),
        "queries_per_s_batch256": 256.0 / (p50b * 1e-3),
        "embedder": f"HashingEmbeddingFunction({args.dim}) on the host (the injectable default; a local transformer embedder runs on the GPU instead)",
    }


def measured_traffic(kernel, dtype, dim, n_local):
    """(bytes per launch, source) — HBM bytes from a COMMITTED rocprofv3 --pmc FETCH_SIZE pass, scaled by rows: counters
    cannot be read from inside this process, so the figure is a replayed constant of the newest profiles/traffic_r*.json
    that knows the kernel, not a measurement of this run.  (None, None) when no counter run exists for the kernel/shape."""
    for name in ("traffic_r3.json", "traffic_r2.json", "traffic_r1.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                t = json.load(f)
            if t["dim"] != dim:
                continue
            return t["bytes_per_row"][kernel][dtype] * n_local, f"replayed from profiles/{name} ({t.get('source', 'rocprofv3 --pmc FETCH_SIZE x2')})"
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def main():
    args = parse_args()
    # stdout carries exactly ONE JSON line: library banners (RCCL prints its version to stdout) go to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    # development rehearsal of the N > 1 code path on a ONE-GPU box (CODD_BENCH_REHEARSAL=gloo): every rank uses cuda:0 and
    # the collective is gloo (RCCL refuses two ranks on one device).  The line says so and is not a measurement.
    rehearsal = os.environ.get("CODD_BENCH_REHEARSAL", "") == "gloo"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from codd_query_engine_amd.knn_index import DeviceKnnIndex
    from codd_query_engine_amd.sharded import ShardedSearcher, shard_bounds

    N, d, B, k = args.rows, args.dim, args.batch, args.k
    lo, hi = shard_bounds(N, world, rank)
    centres = cluster_centres(torch, args, device) if args.corpus == "clustered" else None
    queries = draw(torch, B, d, torch.Generator(device=device).manual_seed(4321), device, centres, args.noise)
    n_planted_q = min(4, B)
    # every timed step searches its own batch (seed 4321 + i; batch 0 carries the planted neighbours): a loop over one
    # batch would re-run the same thresholds, hit counts and cache state K times
    n_batches = max(1, min(args.steps, 16))
    batches = [queries] + [draw(torch, B, d, torch.Generator(device=device).manual_seed(4321 + i), device, centres, args.noise)
                           for i in range(1, n_batches)]

    t_build = time.perf_counter()
    ix = DeviceKnnIndex(d, args.dtype, str(device))
    build_shard(ix, torch, lo, hi, N, d, queries, n_planted_q, k, device, centres, args.noise)
    t_build = time.perf_counter() - t_build
    for kv in args.set:
        key, _, val = kv.partition("=")
        ix.set_option(key, int(val))
    searcher = ShardedSearcher(ix, row_base=lo, always_gather=args.force_dist) if use_dist else None

    def step(q):
        if searcher is not None:
            return searcher.search(q, k)
        return ix.search_tensors(q, k)

    def barrier():
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # correctness gate inside the bench, outside the timed region: (a) the planted neighbours must come back, in order;
    # (b) on one GPU, ALL B queries of two batches must equal the exact scan of the same index (filter switched off) bit
    # for bit — ids and distances: a filter that drops true neighbours of ordinary queries cannot pass.
    # It runs AFTER the timed region (since round 3): its 64 exact scans stream 2 TB from HBM at full clock, and K steps
    # timed right behind them measured a chip that was still shedding that load (--validate-before restores the old order).
    def validate():
        dist_out, rows_out = step(queries)
        torch.cuda.synchronize()
        expect = np.array([[planted_row(b, j, N) for j in range(k)] for b in range(n_planted_q)])
        ok = bool(np.array_equal(rows_out[:n_planted_q].cpu().numpy(), expect))
        # (sharded: every rank switches its filter off for the same two searches, so the comparison is between the merged
        # filtered answer and the merged exact answer)
        for qb in (queries, batches[-1]):
            d_f, r_f = step(qb)
            ix.set_option("filter", 0)
            d_e, r_e = step(qb)
            ix.set_option("filter", 1)
            ok = ok and bool(torch.equal(r_f, r_e) and torch.equal(d_f, d_e))
        torch.cuda.synchronize()
        return ok

    validated_queries = 2 * B
    valid = None
    if args.validate_before:
        valid = validate()
    step(queries)  # (the first search of an index derives the int8 shadow from the rows: ingest work, not a step)
    barrier()
    for i in range(args.warmup):
        step(batches[i % n_batches])
    barrier()

    launches_per_step = 4 * ((B + 255) // 256) + (B + 7) // 8  # upper bound on timed launches per step
    depth = args.pipeline if args.pipeline > 0 else (2 if searcher is not None else 1)

    class _LocalAsync:
        """One GPU, no process group: consecutive batches on `depth` alternating HIP streams (the index keeps one workspace per
        stream), what ShardedSearcher.search_async does for the sharded path."""

        def __init__(self, n):
            self.streams = [torch.cuda.Stream(device=device) for _ in range(n)]
            self.turn = 0

        def search_async(self, q, kk, _depth):
            side = self.streams[self.turn]
            self.turn = (self.turn + 1) % len(self.streams)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                out = ix.search_tensors(q, kk)
            q.record_stream(side)
            from codd_query_engine_amd.sharded import PendingSearch

            return PendingSearch(side, out)

    apipe = searcher if searcher is not None else (_LocalAsync(depth) if depth > 1 else None)
    if depth == 1:
        # one stream: the HIP events around every heavy launch are taken inside the timed region itself
        ix.set_option("profile", args.steps * launches_per_step + 8)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(batches[i % n_batches])
        barrier()
        elapsed = time.perf_counter() - t0
    else:
        # sharded: consecutive batches alternate between `depth` streams, so the small kernels, the all_gather and
        # the merge of batch i overlap the filter kernel of batch i+1.  Every one of the K batches is complete when
        # the closing barrier returns.  Event pairs on concurrent streams would also time the wait for the other
        # stream's kernel, so the per-kernel durations come from K more steps on ONE stream right after.
        for _ in range(depth):
            apipe.search_async(queries, k, depth).result()
        barrier()
        t0 = time.perf_counter()
        pending = []
        for i in range(args.steps):
            pending.append(apipe.search_async(batches[i % n_batches], k, depth))
            if len(pending) >= depth:
                pending.pop(0).result()
        for h in pending:
            h.result()
        barrier()
        elapsed = time.perf_counter() - t0
        ix.set_option("profile", args.steps * launches_per_step + 8)
        for i in range(args.steps):
            step(batches[i % n_batches])
        barrier()
    kernels = {}
    for name in ("scan", "filter", "sample", "finalize"):
        ev = ix.stat(f"events:{name}")
        if ev:
            kernels[name] = {"launches": ev, "avg_ms": ix.stat(f"time_ns:{name}") * 1e-6 / ev}
    ix.set_option("profile", 0)
    ix_shadow8_passes = ix.stat("shadow8_passes")
    ix_tile_passes = ix.stat("i8v2_passes")
    filter_stats = {key: ix.stat(key) for key in ("filter_passes", "fallback_queries", "filter_hits", "filter_survivors")}
    if valid is None:
        valid = validate()

    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    # p50 latency of a single query (B=1), host wall clock incl. D2H of the k results
    lat = []
    q1 = queries[:1].contiguous()
    for i in range(args.latency_iters + 3):
        barrier()
        t1 = time.perf_counter()
        dd, rr = step(q1)
        rr.cpu()
        dd.cpu()
        if i >= 3:
            lat.append((time.perf_counter() - t1) * 1e3)
    p50_ms = statistics.median(lat) if lat else None

    dropin = None
    if world == 1 and not use_dist:
        try:
            dropin = dropin_call(ix, args, torch)
        except Exception as e:  # noqa: BLE001  (never takes the headline number down with it)
            dropin = {"failed": repr(e)}

    elem = 4 if args.dtype == "f32" else 2
    n_local = hi - lo
    ms_per_step = elapsed / args.steps * 1e3
    qps = B * args.steps / elapsed
    # the dominant kernel = the one with the largest share of device time in the timed region
    KERNEL_NAMES = {"scan": "scan_topk_kernel", "filter": "i8_tile_kernel<FILTER>" if ix_tile_passes else "gemm_filter_kernel<FILTER>",
                    "sample": "i8_tile_kernel<SAMPLE>" if ix_tile_passes else "gemm_filter_kernel<SAMPLE>", "finalize": "finalize_kernel"}
    dom = max(kernels, key=lambda kk: kernels[kk]["launches"] * kernels[kk]["avg_ms"]) if kernels else None
    avg_launch_s = kernels[dom]["avg_ms"] * 1e-3 if dom else None
    # algorithmic bytes one launch must stream: the exact scan reads the stored rows once
    # (<= 8 queries ride along); the MFMA filter reads the bf16 shadow of the shard once for 256 queries
    # (a batch of <= 256 queries is filtered through the int8 shadow: 1 B/element, rows padded to 128 elements)
    int8_batch = dom == "filter" and B <= 256 and ix_shadow8_passes > 0
    algo_bytes_launch = n_local * ((d + 127) // 128 * 128) if int8_batch else n_local * d * (2 if dom == "filter" else elem)
    achieved = (algo_bytes_launch / avg_launch_s / 1e9) if avg_launch_s else None
    traffic, traffic_source = measured_traffic("filter8" if int8_batch else dom, args.dtype, d, n_local) if dom else (None, None)
    stored_gbps = (n_local * d * elem / avg_launch_s / 1e9) if avg_launch_s else None
    line = {
        "metric": "queries/sec, top-10 over 10M x 768 corpus",
        "value": qps,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {
            "workload": f"{N} x {d} {args.dtype} corpus, batch={B} queries, top-{k}, exact cosine (BASELINE configs[2])"
                        + (f"; CLUSTERED rows and queries ({args.centres} unit centres + {args.noise} * randn / sqrt(d)): not the headline workload" if centres is not None else ""),
            "rows": N, "dim": d, "batch": B, "k": k,
            "parallelism": f"row-sharded x{world}, one all_gather of B*k u64 per rank" if world > 1 else "single GPU",
            "rows_per_gpu": n_local,
            "batches_in_flight": depth,
            **({"engine_options": args.set} if args.set else {}),
        },
        **({"rehearsal": "all ranks on cuda:0, gloo collective: code-path check, not a measurement"} if rehearsal else {}),
        "p50_latency_ms_batch1": p50_ms,
        "dropin_call": dropin,
        "results_valid": valid,
        # a SELF-comparison: filtered search vs the exact scan of the same engine (the exact scan itself is what the GPU test suite
        # checks against the CPU oracle, tests/test_gpu_parity.py) — not an oracle check inside the bench
        "results_validated": f"{validated_queries} queries (all queries of two batches: ids and distances of the filtered search bit-equal to the EXACT SCAN OF THE SAME "
                             + ("INDEX" if searcher is None else "SHARDS, merged the same way") + " — a self-comparison: the exact scan itself is oracle-checked in the -m gpu suite, "
                             "not in this run; planted neighbours in order)",
        "query_batches": n_batches,
        "index_build_s": t_build,
        "roofline": {
            "bound": "hbm",
            "kernel": KERNEL_NAMES.get(dom) + (" (int8 shadow)" if int8_batch else ""),
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            # frac: on the bytes this kernel MUST move per launch (below); frac_vs_stored_corpus: SURVEY.md §8(d)'s definition,
            # N*d*bytes_per_elem of the STORED rows per launch — above 1 by design when the filter streams a narrower derived
            # copy (int8 shadow: 1 byte per element) and re-scores only survivors from the stored rows
            "frac": (achieved / HBM_PEAK_GBPS) if achieved else None,
            "frac_vs_stored_corpus": (stored_gbps / HBM_PEAK_GBPS) if stored_gbps else None,
            "traffic": traffic,
            "traffic_source": traffic_source,
            "algorithmic_bytes_per_launch": algo_bytes_launch,
            "algorithmic_bytes_are": ("int8 shadow: rows x ceil(dim/128)*128 bytes (one pass serves <= 256 queries)" if int8_batch else
                                      ("bf16 shadow: rows x dim x 2 bytes" if dom == "filter" else f"stored rows: rows x dim x {elem} bytes")),
            "avg_launch_ms": avg_launch_s * 1e3 if avg_launch_s else None,
            "launches_timed": kernels[dom]["launches"] if dom else 0,
            "events_from": "the timed region" if depth == 1 else "the same K steps repeated on one stream after the timed region",
            "all_kernels": kernels,
            # SURVEY.md §8(d) prices a pass at the STORED corpus bytes (N*d*bytes_per_elem); the filter streams a
            # narrower derived copy instead, so against that figure the launch runs above the HBM roof
            "stored_corpus_equivalent_GBps": stored_gbps,
            # the other side of the ridge for the same launch: multiply-accumulates of the filter GEMM (2*B*rows*d per
            # launch) against the dense matrix peak of its operand type (MI355X_MICROARCH.md: bf16 2.5 PFLOP/s, i8 2x that)
            "matrix_side": ({"achieved": 2.0 * B * n_local * d / avg_launch_s / 1e12, "peak": 5000.0 if int8_batch else 2500.0,
                             "unit": "TOP/s" if int8_batch else "TFLOP/s",
                             "frac": 2.0 * B * n_local * d / avg_launch_s / 1e12 / (5000.0 if int8_batch else 2500.0),
                             # what the matrix pipe alone sustains under the chip's power limit on operand bytes like these
                             # (a replayed constant, not measured in this run: profiles/r2/mfma_rate_micro.txt)
                             **({"sustained_peak_measured": 4200.0, "sustained_peak_source": "profiles/r2/mfma_rate_micro.txt (int8 MFMA, Gaussian operand bytes, registers only)"}
                                if int8_batch else {})}
                            if dom == "filter" and avg_launch_s else None),
        },
        "filter_stats": filter_stats,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = cpu_baseline(ix, args, queries.cpu().numpy())
        except Exception as e:  # the baseline must never take the GPU number down with it
            line["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": None, "kind": "port", "sample": f"failed: {e}"}
    sys.stdout.flush()
    if rank == 0:
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    ix.close()
    if use_dist:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if not valid:
        sys.exit(3)


if __name__ == "__main__":
    main()
